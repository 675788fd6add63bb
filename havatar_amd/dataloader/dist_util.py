"""Sampler choice and rank helpers (reference: dataloader/dist_util.py:6-40)."""
import torch.distributed as dist
from torch.utils import data


def data_sampler(dataset, shuffle, distributed):
    if distributed:
        return data.distributed.DistributedSampler(dataset, shuffle=shuffle)
    return data.RandomSampler(dataset) if shuffle else data.SequentialSampler(dataset)


def get_rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def get_world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def synchronize():
    if get_world_size() > 1:
        dist.barrier()
