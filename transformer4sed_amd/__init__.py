"""MI355X-native MAT-SED hot path.

Importing this package changes nothing in the process.  One runtime setting is RECOMMENDED to the entry point that owns the process
(bench.py, a training script) and has to be made before HIP initialises: `GPU_MAX_HW_QUEUES=8` (ROCclr's default is 4).  HIP streams are
dealt round-robin onto that many hardware queues; a train step uses three of its own (main, no-grad teacher pass, weight-gradient stream)
and RCCL adds its streams once a process group exists -- with 4 queues the teacher's stream then lands on the main stream's queue and the
two passes serialise (measured at world size 1 with the RCCL path forced on: +2.4 % -> +0.7 % per step with 8 queues,
profiles/r5_host_contention.txt).  `transformer4sed_amd.hostcpu.recommended_env()` returns it for launch scripts; until round 5 the package
set it as an import side effect."""
