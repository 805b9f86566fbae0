"""bowtie_amd: MI355X-native FM-index search hot path of Bowtie 1.

The package holds only what the path needs: the HIP kernels + C-ABI (csrc/), the ctypes binding
to that ABI (_lib.py, aligner.py) and the host-side read-in / hit-out surface (reads.py,
output.py).  Importing the package does not load the native library; `bowtie_amd.aligner` does,
and fails loudly if it is missing.
"""
__version__ = "0.1.0"
