"""Command-line entry of the mapper drop-in.

Accepts exactly the flags phASER passes to its mapper stage (phaser/phaser.py:1346 builds the command line;
phaser/call_read_variant_map.py:18-22 declares them) and forwards them to the HIP-backed
`phaser_amd.read_variant_map.do_read_variant_map`.  SAM text is read from stdin.

    samtools view -h ... | python3 -m phaser_amd.call_read_variant_map --variant_table T --baseq 10 --o OUT
"""
import argparse
import sys

# flag -> (type, default, required); kept as data so the help text and the forwarding below cannot drift apart
MAPPER_FLAGS = (
    ("variant_table", str, None, True),
    ("baseq", int, 10, False),
    ("o", str, None, True),
    ("splice", int, 1, False),
    ("isize_cutoff", float, 0, False),
)


def parse(argv=None) -> argparse.Namespace:
    ap = argparse.ArgumentParser(description="phASER read->variant mapper (MI355X build)")
    for flag, kind, default, needed in MAPPER_FLAGS:
        if needed:
            ap.add_argument("--" + flag, type=kind, required=True)
        else:
            ap.add_argument("--" + flag, type=kind, default=default)
    return ap.parse_args(argv)


def main(argv=None) -> int:
    from .read_variant_map import do_read_variant_map
    ns = parse(argv)
    do_read_variant_map(*(getattr(ns, flag) for flag, _, _, _ in MAPPER_FLAGS))
    return 0


if __name__ == "__main__":
    sys.exit(main())
