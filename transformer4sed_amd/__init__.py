"""MI355X-native MAT-SED hot path.

One process-wide runtime setting is made here, before HIP initialises: `GPU_MAX_HW_QUEUES` (default 4 in ROCclr) -> 8, unless the
environment already says otherwise.  HIP streams are dealt round-robin onto that many hardware queues; a train step uses three of its
own (main, no-grad teacher pass, weight-gradient stream) and RCCL adds its streams once a process group exists -- with 4 queues the
teacher's stream then lands on the main stream's queue and the two passes serialise (measured at world size 1 with the RCCL path forced
on: +2.4 % -> +0.7 % per step with 8 queues, profiles/r5_host_contention.txt)."""
import os

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
