"""phaser_amd: MI355X-native read-backed phasing hot path (phASER drop-in for the mapper + phasing core)."""
import os as _os

# The ROCm runtime spreads a process's streams over GPU_MAX_HW_QUEUES hardware queues (default 4) and reads the variable when it starts.  The device BAM
# decoder keeps eight copy streams and up to three K_inflate streams busy at once; streams that share a queue wait for each other (a copy behind a 60 ms
# kernel), so the package asks for 16 queues -- unless the user has set the variable -- before anything touches the GPU.  A host application that has
# already started the runtime keeps what it has; the decoder then inflates with one launch per 4 GB instead (phz_bamdev.hip).
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

__version__ = "0.1.0"
