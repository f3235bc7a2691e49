"""ofps_amd -- MI355X-native (gfx950) backend for the OFPS flow hot path.

csrc/        hand-written HIP kernels + the C ABI of include/ofps_hip.h  (-> libofps_hip.so, built by ofps_amd.build)
_lib.py      ctypes prototypes of that ABI; load() fails loudly when the library or a symbol is missing
runtime.py   HipContext: thin Python wrapper of one ofps_hip_ctx (what the tests and bench.py call)
plugins.py   Python mirror of the reference's Decoder / Detector / Estimator / Properties surface
host/        the same mirror in C++ plus the detection / tracking loops and a CLI
mvec.py      the reference's .mvec interchange format
synth.py     seeded synthetic inputs
distributed.py  frame-pair sharding helpers for one-process-per-GPU runs

There is no CPU fallback in this package and nothing in it imports oracle/ (the CPU restatement is test infrastructure).
"""

__version__ = "0.1.0"
