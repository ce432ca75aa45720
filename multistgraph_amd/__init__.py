"""MI355X-native Multi-ATGCN hot path (see README.md / DESIGN.md)."""
import os as _os

# HIP maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  The path runs the layers' chains, the
# hoisted x parts and the backward's side work on streams of its own; once an RCCL communicator holds queues too (any
# torch.distributed job on the "nccl" backend), two of the chains end up sharing a queue and serialise: forward 8.0 instead
# of 6.8 ms, training step 25.2 instead of 22.4 ms at the headline shape, every kernel's own duration unchanged (profiles/r04_rccl_queues_lab.log; 8 is the best
# setting measured, 10 and more cost the training step 20 %).  The runtime reads the variable when it starts, so this only
# helps when the package is imported before the first HIP call of the process - otherwise export it in the job's
# environment.  A value that is already set is left alone.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
