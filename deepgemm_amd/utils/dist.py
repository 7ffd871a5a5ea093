"""Process-group bootstrap for the expert-parallel harness (one process per GPU, RCCL over xGMI).

Same helper names as the reference's test bootstrap (deep_gemm/utils/dist.py:10-74); backend 'nccl' is RCCL on ROCm and
'gloo' is used by the CPU tests.  Rendezvous is on 127.0.0.1 unless MASTER_ADDR says otherwise."""
import os
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def init_dist(local_rank: int, num_local_ranks: int, backend: Optional[str] = None) -> Tuple[int, int, dist.ProcessGroup]:
    ip = os.environ.get('MASTER_ADDR', '127.0.0.1')
    port = int(os.environ.get('MASTER_PORT', '8361'))
    num_nodes = int(os.environ.get('WORLD_SIZE', 1)) if 'LOCAL_WORLD_SIZE' not in os.environ else 1
    node_rank = int(os.environ.get('RANK', 0)) if 'LOCAL_WORLD_SIZE' not in os.environ else 0
    if backend is None:
        backend = 'nccl' if torch.cuda.is_available() else 'gloo'
    if backend == 'nccl':
        torch.cuda.set_device(local_rank)
    if not dist.is_initialized():
        dist.init_process_group(backend=backend, init_method=f'tcp://{ip}:{port}',
                                world_size=num_nodes * num_local_ranks, rank=node_rank * num_local_ranks + local_rank)
    return dist.get_rank(), dist.get_world_size(), dist.new_group(list(range(dist.get_world_size())))


def uneven_all_gather(tensor: torch.Tensor, dim: int = 0, group: Optional[dist.ProcessGroup] = None) -> torch.Tensor:
    """All-gather of tensors whose size along ``dim`` differs per rank (sizes exchanged first, then padded gather)."""
    world = dist.get_world_size(group)
    size = torch.tensor([tensor.size(dim)], dtype=torch.long, device=tensor.device)
    sizes = [torch.empty_like(size) for _ in range(world)]
    dist.all_gather(sizes, size, group=group)
    sizes = [int(s.item()) for s in sizes]
    longest = max(sizes)
    pad_shape = list(tensor.shape)
    pad_shape[dim] = longest
    padded = tensor.new_zeros(pad_shape)
    padded.narrow(dim, 0, tensor.size(dim)).copy_(tensor)
    gathered: List[torch.Tensor] = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(gathered, padded, group=group)
    return torch.cat([g.narrow(dim, 0, s) for g, s in zip(gathered, sizes)], dim=dim)


def dist_print(s: str = '', once_in_node: bool = False) -> None:
    if not once_in_node or dist.get_rank() % max(torch.cuda.device_count(), 1) == 0:
        print(s, flush=True)
    dist.barrier()
