"""MI355X-native Multi-ATGCN hot path (see README.md / DESIGN.md).

Data-parallel jobs (torch.distributed on the "nccl" = RCCL backend) should export GPU_MAX_HW_QUEUES=8 before the
process starts: INTEGRATION.md section 5, profiles/r04_rccl_queues_lab.log."""
