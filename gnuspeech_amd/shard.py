"""Multi-GPU sharding of an utterance batch: voices are independent, so rank r simply owns a
contiguous range of the (length-sorted) voice index and no data-path collective exists.  The only
communication is the timing barrier / max-over-ranks of the benchmark contract."""
import ctypes as C

import numpy as np

from ._capi import TrmDerived, check, lib


def shard_range(nvoices, rank, world):
    """Contiguous, balanced, disjoint ranges covering [0, nvoices): the first nvoices % world ranks get one more."""
    base, extra = divmod(int(nvoices), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def sort_by_length(nframes):
    """Order that groups voices of similar length into the same 64-voice workgroup."""
    return np.argsort(np.asarray(nframes), kind="stable")[::-1]


def samples_for_frames(inputParameters, nframes):
    """Output samples per voice, without a device (TRMSampleRateConverter bookkeeping)."""
    return lib().trm_samples_for_frames(C.byref(inputParameters.c), int(nframes))


def derive(inputParameters):
    d = TrmDerived()
    check(lib().trm_derive(C.byref(inputParameters.c), C.byref(d)))
    return {k: getattr(d, k) for k, _ in TrmDerived._fields_}


def max_over_ranks(seconds, device=None):
    """The benchmark contract's timing reduction: MAX over ranks (identity without a process group)."""
    import torch
    import torch.distributed as td
    if not (td.is_available() and td.is_initialized()):
        return float(seconds)
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    td.all_reduce(t, op=td.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device=None):
    import torch
    import torch.distributed as td
    if not (td.is_available() and td.is_initialized()):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    td.all_reduce(t, op=td.ReduceOp.SUM)
    return float(t.item())
