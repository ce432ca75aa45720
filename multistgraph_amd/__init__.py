"""MI355X-native Multi-ATGCN hot path (see README.md / DESIGN.md).

Data-parallel jobs (torch.distributed on the "nccl" = RCCL backend) call sharding.use_own_stream_pool() before their first
forward: INTEGRATION.md section 5, profiles/r04_rccl_queues_lab.log."""
