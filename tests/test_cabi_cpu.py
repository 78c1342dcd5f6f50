"""C-ABI checks that need no GPU: the shared library loads, exports every symbol include/iefvad.h
declares, and the ctypes structures have the layout a C compiler gives the header."""
import ctypes as C
import os
import re
import subprocess

import pytest

import iefvad_amd
from iefvad_amd import lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "iefvad.h")


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(L.LIB_PATH):
        L.build_library()
    return L.load_library()


def test_exports_every_declared_symbol(lib):
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    declared = sorted(set(re.findall(r"\b(iefvad_[a-z0-9_]+)\s*\(", text)))
    assert declared == sorted(L.SYMBOLS)
    for s in declared:
        assert hasattr(lib, s), s
    assert lib.iefvad_abi_version() == L.ABI_VERSION


def test_struct_layout_matches_c_compiler(tmp_path):
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "iefvad.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n",'
                   'sizeof(iefvad_config),sizeof(iefvad_weights),sizeof(iefvad_outputs),sizeof(iefvad_stage_times),'
                   'offsetof(iefvad_weights,ref_w1),offsetof(iefvad_weights,cls_w),offsetof(iefvad_config,lambda_ref),'
                   'sizeof(iefvad_train_options),offsetof(iefvad_train_options,seed),offsetof(iefvad_train_options,keep_mask),'
                   'sizeof(iefvad_output_grads),sizeof(iefvad_weight_grads),sizeof(iefvad_unit_io),offsetof(iefvad_unit_io,w),offsetof(iefvad_unit_io,logits),sizeof(iefvad_adamw_tensor),offsetof(iefvad_adamw_tensor,first_chunk));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    want = [C.sizeof(L.Config), C.sizeof(L.Weights), C.sizeof(L.Outputs), C.sizeof(L.StageTimes),
            L.Weights.ref_w1.offset, L.Weights.cls_w.offset, L.Config.lambda_ref.offset,
            C.sizeof(L.TrainOptions), L.TrainOptions.seed.offset, L.TrainOptions.keep_mask.offset, C.sizeof(L.OutputGrads),
            C.sizeof(L.WeightGrads), C.sizeof(L.UnitIO), L.UnitIO.w.offset, L.UnitIO.logits.offset, 48, 40]      # iefvad_adamw_tensor: six 8-byte fields (losses.AdamW builds it as a [count, 6] uint64 table)
    assert got == want


def test_create_rejects_bad_config_without_touching_the_gpu(lib):
    h = C.c_void_p()
    base = dict(abi_version=L.ABI_VERSION, embed_dim=768, seq_len=256, num_heads=8, num_layers=2, num_steps=10,
                noise_model=1, compute=0, lambda_ref=0.5, nu=8.0, epsilon=1e-8, micro_batch=0)
    for bad, frag in [(dict(embed_dim=512), "D=768"), (dict(num_layers=0), "num_layers"), (dict(num_steps=65), "num_steps"),
                      (dict(noise_model=7), "Unsupported noise_model"), (dict(abi_version=9), "abi_version"),
                      (dict(compute=5), "compute mode")]:
        cfg = L.Config(**dict(base, **bad))
        assert lib.iefvad_create(C.byref(cfg), C.byref(h)) != 0
        assert frag in L.last_error(), (bad, L.last_error())
        assert not h.value
    assert lib.iefvad_workspace_bytes(None, 4) == 0


def test_gather_entry_points_reject_bad_arguments_without_a_gpu(lib):
    """iefvad_comm_* / iefvad_gather_scores validate before they touch RCCL or the device."""
    h = C.c_void_p()
    ident = C.create_string_buffer(L.COMM_ID_BYTES)
    assert lib.iefvad_comm_create(ident, 2, 2, C.byref(h)) != 0 and "rank 2 of 2" in L.last_error()
    assert lib.iefvad_comm_create(None, 1, 0, C.byref(h)) != 0 and "null" in L.last_error()
    assert lib.iefvad_comm_unique_id(None) != 0
    assert lib.iefvad_comm_nranks(None) == 0
    assert lib.iefvad_gather_scores(None, None, 0, None, None, 0, None) != 0 and "null" in L.last_error()
    lib.iefvad_comm_destroy(None)


def _plan(lib, nranks, rank, counts=None, count=0):
    carr = (C.c_int64 * nranks)(*counts) if counts is not None else None
    summary = (C.c_int64 * 5)()
    steps = (C.c_int64 * (4 * max(nranks - 1, 1)))()
    assert lib.iefvad_gather_plan(nranks, rank, carr, count, summary, steps) == 0, L.last_error()
    n = int(summary[4])
    return dict(equal=bool(summary[0]), my_offset=int(summary[1]), my_count=int(summary[2]), total=int(summary[3]),
                steps=[tuple(int(steps[4 * i + j]) for j in range(4)) for i in range(n)])


def test_gather_plan_is_the_ordered_exchange(lib):
    """The exchange `iefvad_gather_scores` enqueues, as data (`iefvad_gather_plan`: the same `gather_plan()` the gather runs
    through), for the shard shapes `harness.partition_by_snippets` produces -- unequal counts, zero-count ranks, one rank --
    checked on the host: no GPU and no second rank needed.  The N > 1 RCCL TRANSPORT itself has not run on hardware (a test
    box has one GPU); what is pinned here is every (peer, offset, count) it would be given.  A simulated exchange in numpy
    must reproduce rank-order concatenation (= the reference's sequential order, test.py:123-129,153) on every rank."""
    import numpy as np
    for counts in ([5, 0, 7, 3], [0, 0, 4], [1000, 1000, 1000], [9], [0, 0], [3, 3, 0, 3, 1, 2, 8, 5]):
        n = len(counts)
        data = [np.arange(c, dtype=np.float32) + 1000 * r for r, c in enumerate(counts)]
        want = np.concatenate(data) if sum(counts) else np.zeros(0, np.float32)
        plans = [_plan(lib, n, r, counts) for r in range(n)]
        offs = np.concatenate([[0], np.cumsum(counts)])
        for r, p in enumerate(plans):
            assert p["total"] == sum(counts) and p["my_count"] == counts[r] and p["my_offset"] == offs[r]
            assert p["equal"] == (len(set(counts)) == 1)
            got = np.full(sum(counts), -1.0, np.float32)
            if p["equal"]:
                assert p["steps"] == []
                got = want.copy()                                  # one ncclAllGather: rank order by definition
            else:
                assert [s[0] for s in p["steps"]] == [q for q in range(n) if q != r]       # every peer once, rank order
                for peer, send, roff, rcount in p["steps"]:
                    assert send == counts[r] and rcount == counts[peer] and roff == offs[peer]
                    # the peer's matching step sends exactly what this rank expects to receive
                    back = [s for s in plans[peer]["steps"] if s[0] == r][0]
                    assert back[1] == rcount and back[3] == send
                    got[roff:roff + rcount] = data[peer]
                got[p["my_offset"]:p["my_offset"] + counts[r]] = data[r]
            assert np.array_equal(got, want), (counts, r)
    # counts == NULL: every rank contributes `count`
    p = _plan(lib, 4, 2, None, 250)
    assert p == dict(equal=True, my_offset=500, my_count=250, total=1000, steps=[])
    bad = (C.c_int64 * 2)(4, -1)
    summary = (C.c_int64 * 5)()
    assert lib.iefvad_gather_plan(2, 0, bad, 0, summary, None) != 0 and "negative count" in L.last_error()
    assert lib.iefvad_gather_plan(2, 2, None, 1, summary, None) != 0 and "rank" in L.last_error()


def test_shim_refuses_cpu_tensors():
    import argparse
    import torch
    args = argparse.Namespace(visual_layers=1, visual_head=8, num_refinement_steps=0, lambda_ref=0.5,
                              noise_model="StudentT", nu=8)
    m = iefvad_amd.MMFMIL(14, 768, 256, 768, 8, 1, 8, 10, 10, "cuda", args).eval()
    x = torch.zeros(1, 256, 768)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(x, x, None, None, None)


def test_host_gather_concatenates_rows_without_a_gpu(lib):
    """`iefvad_host_gather` (the loader side of iefvad_forward_videos): pieces of uneven size, empty pieces, one or several
    threads, small and large totals -- the destination is the plain concatenation."""
    import numpy as np
    rng = np.random.default_rng(0)
    for sizes, threads in (([5, 0, 7, 1], 4), ([3 << 20, 11, 0, 2 << 20, 5 << 20, 1 << 20, 123457], 4), ([1 << 22] * 5, 16), ([9], 0)):
        parts = [rng.integers(0, 255, n, dtype=np.uint8) for n in sizes]
        dst = np.full(sum(sizes) + 8, 0xEE, np.uint8)
        ptrs = (C.c_void_p * len(parts))(*[p.ctypes.data for p in parts])
        nb = (C.c_size_t * len(parts))(*sizes)
        assert lib.iefvad_host_gather(dst.ctypes.data, ptrs, nb, len(parts), threads) == 0, L.last_error()
        assert np.array_equal(dst[:-8], np.concatenate(parts)) and (dst[-8:] == 0xEE).all()
    # misaligned sources and destination (the copy uses 32-byte non-temporal stores behind an alignment head)
    base = rng.integers(0, 255, (1 << 20) + 64, dtype=np.uint8)
    parts = [base[3:3 + 70001], base[17:17 + 4097], base[1:1 + (1 << 19)]]
    dst = np.full(sum(p.size for p in parts) + 13 + 8, 0xEE, np.uint8)
    ptrs = (C.c_void_p * 3)(*[p.ctypes.data for p in parts])
    nb = (C.c_size_t * 3)(*[p.size for p in parts])
    assert lib.iefvad_host_gather(dst.ctypes.data + 13, ptrs, nb, 3, 2) == 0, L.last_error()
    assert np.array_equal(dst[13:-8], np.concatenate(parts)) and (dst[-8:] == 0xEE).all() and (dst[:13] == 0xEE).all()
    assert lib.iefvad_host_gather(None, None, None, 0, 4) == 0
    assert lib.iefvad_host_gather(None, None, None, 3, 4) != 0 and "null" in L.last_error()


def test_host_gather_bf16_rounds_as_torch_does(lib):
    """`iefvad_host_gather_bf16` -- what the list walk's copy threads run for wire_dtype = BF16: fp32 pieces in, one bf16 stream out,
    bit for bit `torch.Tensor.to(torch.bfloat16)` (round to nearest even) on random bit patterns of every exponent, ties, values
    that round up to the next binade or to +-inf, denormals, zeros of both signs and infinities; NaNs stay NaNs with their sign.
    One and many threads (byte ranges that start inside a piece), pieces of uneven size, aligned and misaligned buffers."""
    import numpy as np
    import torch
    rng = np.random.default_rng(1)
    special = np.array([0x00000000, 0x80000000, 0x7F800000, 0xFF800000, 0x7F7FFFFF, 0xFF7FFFFF, 0x7F7F8000, 0x7F7F7FFF, 0x3F808000,
                        0x3F818000, 0x3F80FFFF, 0x3F7FFFFF, 0x00000001, 0x007FFFFF, 0x00008000, 0x00018000, 0x7FC00000, 0xFFC00001,
                        0x7F800001, 0xFF8ABCDE, 0x7FFFFFFF, 0x3F800000], dtype=np.uint32)
    for sizes, threads, shift in (([16], 1, 0), ([768 * 37, 768, 768 * 300, 768 * 5], 1, 0), ([768 * 701, 768 * 3, 768 * 1500, 768 * 256], 7, 0),
                                  ([768 * 400] * 6, 16, 0), ([768 * 90, 768 * 411], 3, 1)):
        parts = []
        for n in sizes:
            bits = rng.integers(0, 1 << 32, n + shift, dtype=np.uint64).astype(np.uint32)
            bits[shift:shift + min(n, special.size)] = special[:min(n, special.size)]
            parts.append(bits.view(np.float32)[shift:])                       # shift = 1: sources at 4 mod 32
        total = sum(sizes)
        raw = np.full(total + 16 + 16, 0xEEEE, np.uint16)
        dst = raw[16 * shift:]                                                # shift = 1: destination 32 bytes on (still 32-byte aligned)
        ptrs = (C.c_void_p * len(parts))(*[p.ctypes.data for p in parts])
        nb = (C.c_size_t * len(parts))(*[4 * n for n in sizes])
        assert lib.iefvad_host_gather_bf16(dst.ctypes.data, ptrs, nb, len(parts), threads) == 0, L.last_error()
        src = np.concatenate(parts)
        want = torch.from_numpy(src.copy()).to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16)
        got = dst[:total]
        nan = np.isnan(src)
        assert np.array_equal(got[~nan], want[~nan]), (sizes, threads)
        assert ((got[nan] & 0x7F80) == 0x7F80).all() and ((got[nan] & 0x007F) != 0).all()            # still NaN ...
        assert np.array_equal(got[nan] >> 15, src[nan].view(np.uint32) >> 31)                         # ... of the same sign
        assert (dst[total:total + 16] == 0xEEEE).all()
    # a destination that is not 32-byte aligned takes the scalar path: same bits
    src = rng.standard_normal(768 * 3).astype(np.float32)
    raw = np.zeros(768 * 3 + 32, np.uint16)
    off = (-(raw.ctypes.data // 2) % 16) + 1                                  # 2 bytes past a 32-byte boundary
    ptrs = (C.c_void_p * 1)(src.ctypes.data)
    nb = (C.c_size_t * 1)(src.nbytes)
    assert lib.iefvad_host_gather_bf16(raw.ctypes.data + 2 * off, ptrs, nb, 1, 2) == 0, L.last_error()
    assert np.array_equal(raw[off:off + src.size], torch.from_numpy(src).to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16))
    nb = (C.c_size_t * 1)(100)
    assert lib.iefvad_host_gather_bf16(raw.ctypes.data, ptrs, nb, 1, 2) != 0 and "multiple of 64" in L.last_error()
    assert lib.iefvad_host_gather_bf16(None, None, None, 2, 2) != 0 and "null" in L.last_error()


def test_host_gather_code_is_clean_under_thread_and_address_sanitizers(tmp_path):
    """tests/cabi/Makefile: ief-vad_amd/csrc/hostgather.h (GatherPool, iefvad_host_gather, iefvad_host_gather_bf16 -- the host
    threading code of the whole-video path, the very header libiefvad.so is built from) compiled by gcc with -fsanitize=thread and
    with -fsanitize=address,undefined into a CPU-only stress driver: the entry points on ragged / empty / unaligned pieces with
    1..16 threads, and 1,000 jobs on one persistent pool that GROWS between jobs while the previous job's tables are already freed
    (a thread created by a later start() must not run an earlier job).  Any sanitizer report fails the run."""
    import shutil
    import subprocess
    if shutil.which("g++") is None or shutil.which("make") is None:
        pytest.skip("g++ / make not available")
    cabi = os.path.join(ROOT, "tests", "cabi")
    r = subprocess.run(["make", "-C", cabi, f"OUT={tmp_path}", "check"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count("hostgather_san: all checks passed") == 2
