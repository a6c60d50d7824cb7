"""CLI shim with the reference's flags (phaser/call_read_variant_map.py:14-26)."""
import argparse

from . import read_variant_map


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("--variant_table", type=str, required=True)
    parser.add_argument("--baseq", type=int, default=10)
    parser.add_argument("--o", type=str, required=True)
    parser.add_argument("--splice", type=int, default=1)
    parser.add_argument("--isize_cutoff", type=float, default=0)
    args = parser.parse_args()
    read_variant_map.do_read_variant_map(args.variant_table, args.baseq, args.o, args.splice, args.isize_cutoff)


if __name__ == "__main__":
    main()
