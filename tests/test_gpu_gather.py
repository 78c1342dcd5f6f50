"""`iefvad_gather_scores` (include/iefvad.h) on the one GPU a test box has: an RCCL communicator of
ONE rank exercises the dlopen binding of librccl, ncclCommInitRank / ncclCommCount, the equal-count ncclAllGather
path and the unequal-count bookkeeping (which at one rank reduces to the device-to-device copy of the own slice).
Multi-rank ordering is covered on gloo by tests/test_distributed_cpu.py and tests/test_bench_launcher_cpu.py; the
round driver runs the real N > 1 job.  The reference only fixes the ORDER of the score vector
(/root/reference/test.py:123-129,153).  `-m gpu`."""
import ctypes as C
import os
import socket

import pytest
import torch

from iefvad_amd import harness
from iefvad_amd import lib as L

pytestmark = pytest.mark.gpu


def test_c_abi_gather_world_of_one():
    lib = L.load_library()
    ident = C.create_string_buffer(L.COMM_ID_BYTES)
    assert lib.iefvad_comm_unique_id(ident) == 0, L.last_error()
    h = C.c_void_p()
    torch.cuda.set_device(0)
    assert lib.iefvad_comm_create(ident, 1, 0, C.byref(h)) == 0, L.last_error()
    assert lib.iefvad_comm_nranks(h) == 1
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    local = torch.arange(4096, dtype=torch.float32, device="cuda") * 0.5
    out = torch.full((4096,), -1.0, device="cuda")
    assert lib.iefvad_gather_scores(h, local.data_ptr(), 4096, None, out.data_ptr(), 4096, st) == 0, L.last_error()
    torch.cuda.synchronize()
    assert torch.equal(out, local)
    counts = (C.c_int64 * 1)(1000)
    out2 = torch.full((1000,), -1.0, device="cuda")
    assert lib.iefvad_gather_scores(h, local.data_ptr(), 0, counts, out2.data_ptr(), 1000, st) == 0, L.last_error()
    torch.cuda.synchronize()
    assert torch.equal(out2, local[:1000])
    bad = (C.c_int64 * 1)(-3)
    assert lib.iefvad_gather_scores(h, local.data_ptr(), 0, bad, out2.data_ptr(), 1000, st) != 0 and "negative count" in L.last_error()
    # capacity and overlap are checked before anything is enqueued
    assert lib.iefvad_gather_scores(h, local.data_ptr(), 4096, None, out.data_ptr(), 4095, st) != 0 and "holds 4095" in L.last_error()
    assert lib.iefvad_gather_scores(h, local.data_ptr(), 64, None, local.data_ptr() + 16, 64, st) != 0 and "overlaps" in L.last_error()
    # in place (local IS its own slot) is what ncclAllGather allows
    same = local.clone()
    assert lib.iefvad_gather_scores(h, same.data_ptr(), 4096, None, same.data_ptr(), 4096, st) == 0, L.last_error()
    torch.cuda.synchronize()
    assert torch.equal(same, local)
    assert lib.iefvad_rccl_version() > 0
    lib.iefvad_comm_destroy(h)


def test_harness_gather_goes_through_the_library_on_an_rccl_group():
    import torch.distributed as dist
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        local = torch.rand(70000, device="cuda")
        full = harness.gather_scores(local, counts=[70000])
        torch.cuda.synchronize()
        assert torch.equal(full, local)
        (comm,) = harness._score_comms.values()
        assert comm.nranks == 1 and comm.world == 1
        full2 = harness.gather_scores(local)                      # counts exchanged through torch.distributed first
        assert torch.equal(full2, local)
        comm.close()
        harness._score_comms.clear()
    finally:
        dist.destroy_process_group()
