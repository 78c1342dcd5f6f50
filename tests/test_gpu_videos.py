"""`iefvad_forward_videos` (include/iefvad.h, csrc/ragged.h): whole videos cross the boundary as their VALID rows; the
chunker (tools.py:100-114), the conditional nan_to_num (test.py:90-95) and the `[0:len]` slicing (test.py:119-121,131-138)
run on the device; the encoder keeps ONE pad row per chunk (all pad rows of a window are identical in every layer, the
attention kernels read row min(r, valid)) and everything behind it runs on that row set.  Checked against the reference-shaped
route: host-side process_split + `_unpack_item` + the dense forward on zero-padded chunks, which the capture tests pin to the
reference's own test()."""
import argparse

import numpy as np
import pytest
import torch

import iefvad_amd
from iefvad_amd import harness, synth
from oracle import iefvad_oracle as orc
from tests import helpers as H

pytestmark = pytest.mark.gpu

EDGE_LENGTHS = [37, 255, 256, 257, 512, 1500, 1, 300, 64, 768]      # SURVEY 8d config 1's chunk-edge cases and then some


def make_model(compute, L=2, K=3, **kw):
    sd = synth.make_state_dict(41, 768, L, K)
    args = argparse.Namespace(visual_layers=L, visual_head=8, num_refinement_steps=K, lambda_ref=0.5, noise_model="StudentT", nu=8)
    m = iefvad_amd.MMFMIL(14, 768, 256, 768, 8, L, 8, 10, 10, "cuda", args, compute=compute, **kw)
    m.load_state_dict(sd)
    return m.to("cuda:0").eval(), sd


def videos(lengths, seed=6, dtype=np.float32):
    return [synth.make_video(seed, i, int(n), dtype=dtype) for i, n in enumerate(lengths)]


def dense_reference(model, vids):
    """The reference-shaped route: zero-padded chunks of every video in one dense forward, then [0:len] per video."""
    ci = [harness.process_split(v[0], 256)[0].reshape(-1, 256, 768) for v in vids]
    ce = [harness.process_split(v[1], 256)[0].reshape(-1, 256, 768) for v in vids]
    img = torch.from_numpy(np.concatenate(ci)).cuda()
    ev = torch.from_numpy(np.concatenate(ce)).cuda()
    with torch.no_grad():
        out = model(img, ev, None, None, None)
    lg, wi, we = out["logits"].reshape(-1), out["w_i_mean"].reshape(-1), out["w_e_mean"].reshape(-1)
    res, off = {"logits": [], "w_i_mean": [], "w_e_mean": []}, 0
    for v, c in zip(vids, ci):
        n = v[0].shape[0]
        for k, t in (("logits", lg), ("w_i_mean", wi), ("w_e_mean", we)):
            res[k].append(t[off:off + n])
        off += c.shape[0] * 256
    return {k: torch.cat(v) for k, v in res.items()}


def ragged(model, vids, **kw):
    img = torch.from_numpy(np.concatenate([v[0] for v in vids])).cuda()
    ev = torch.from_numpy(np.concatenate([v[1] for v in vids])).cuda()
    with torch.no_grad():
        return model.forward_videos(img, ev, [v[0].shape[0] for v in vids], **kw)


@pytest.mark.parametrize("compute,micro_batch", [("f32", 0), ("f32", 3), ("bf16", 0), ("bf16", 5), ("fp16x3", 0)])
def test_forward_videos_is_bit_identical_to_the_padded_forward(compute, micro_batch):
    """Chunk-edge lengths (len < 256, == 255 / 256 / 257, multiples of 256, a one-snippet video, a six-chunk video); with a
    small micro-batch the call runs as several passes, each compacting its own valid rows."""
    model, _ = make_model(compute, outputs="scores", micro_batch=micro_batch)
    vids = videos(EDGE_LENGTHS)
    want = dense_reference(model, vids)
    got = ragged(model, vids)
    for k in want:
        assert got[k].shape == want[k].shape == (sum(EDGE_LENGTHS),)
        if compute == "bf16" and k != "logits":
            # the fused heads + fusion kernel (full grids) and the stand-alone fusion kernel sum a row's 768 weights in
            # different orders; which one runs depends on the row count, and the tail now has fewer rows
            assert float((got[k] - want[k]).abs().max()) <= 1e-6, k
        else:
            assert torch.equal(got[k], want[k]), (k, (got[k] - want[k]).abs().max().item())


@pytest.mark.parametrize("compute", ["f32", "bf16"])
def test_whole_chunk_encoder_switch_gives_the_same_bits(compute, monkeypatch):
    """IEFVAD_DENSE_ENCODER=1 (read at model creation) keeps whole 256-row chunks in the encoder and gathers the valid rows
    behind it -- round 3's first design, kept for the A/B: same scores, bit for bit."""
    vids = videos(EDGE_LENGTHS, seed=12)
    model, _ = make_model(compute, outputs="scores")
    got = ragged(model, vids)
    del model
    monkeypatch.setenv("IEFVAD_DENSE_ENCODER", "1")
    model, _ = make_model(compute, outputs="scores")
    monkeypatch.delenv("IEFVAD_DENSE_ENCODER")
    want = ragged(model, vids)
    assert torch.equal(got["logits"], want["logits"])
    assert float((got["w_i_mean"] - want["w_i_mean"]).abs().max()) <= (1e-6 if compute == "bf16" else 0.0)


def test_forward_videos_large_batch_bf16_kernels_and_bf16x6_tolerance():
    """A list large enough for the full-grid bf16 kernels on BOTH sides of the compaction (fused out_proj + LayerNorm, fused
    heads + fusion, the refinement chain): still bit-identical, although the tail now runs on ~60 % of the rows.  bf16x6: the
    compact tail may run on other tilings than the dense one (same arithmetic, fp32-accurate): fp32 gates."""
    lengths = synth.lognormal_lengths(8, 420, 60000, lo=16, hi=3000)
    vids = videos(lengths, seed=8)
    model, sd = make_model("bf16", K=10, outputs="scores")
    want, got = dense_reference(model, vids), ragged(model, vids)
    assert torch.equal(got["logits"], want["logits"])
    for k in ("w_i_mean", "w_e_mean"):
        assert torch.equal(got[k], want[k]), k        # both sides run the fused heads + fusion kernel here
    del model
    model, _ = make_model("bf16x6", K=10, outputs="scores")
    want, got = dense_reference(model, vids), ragged(model, vids)
    assert float((torch.sigmoid(got["logits"]) - torch.sigmoid(want["logits"])).abs().max()) <= H.TOL_SIGMOID
    assert float((got["logits"] - want["logits"]).abs().max()) <= H.TOL_LOGIT
    assert float((got["w_i_mean"] - want["w_i_mean"]).abs().max()) <= 1e-5


@pytest.mark.parametrize("dtype", [np.float32, np.float16])
def test_forward_videos_nan_rule_matches_the_host_rule(dtype):
    """test.py:90-95: `if torch.isnan(x).any(): x = torch.nan_to_num(x, nan=0.0)` per video and modality -- NaN -> 0 and, only in
    that case, +-inf -> the dtype's max / min.  Video 1: NaN + inf in the image features; video 3: inf only (no replacement: the
    chunk's scores go NaN, as in the reference); video 4: NaN in the event features of its second chunk.  Compared with the
    host rule (`harness._unpack_item`) followed by the dense forward, bit for bit including the NaN pattern."""
    lengths = [100, 300, 50, 80, 600, 256]
    vids = videos(lengths, seed=9, dtype=dtype)
    vids[1][0][7, 5] = np.nan
    vids[1][0][290, 100] = np.inf
    vids[1][0][3, 9] = -np.inf
    vids[3][0][10, 10] = np.inf
    vids[4][1][400, 767] = np.nan
    model, _ = make_model("f32", outputs="scores")

    def items():
        for img, ev in vids:
            ci, n = harness.process_split(img, 256)
            ce, _ = harness.process_split(ev, 256)
            yield torch.tensor(ci).unsqueeze(0), torch.tensor(ce).unsqueeze(0), ("Normal",), torch.tensor([n])

    host, _, wi_h, _ = harness.score_loader(model, items(), 256, "cuda:0", "ucfcrime", batch_chunks=4, ragged=False)
    dev, _, wi_d, _ = harness.score_loader(model, items(), 256, "cuda:0", "ucfcrime", batch_chunks=4, ragged=True)
    for i, (a, b) in enumerate(zip(host, dev)):
        assert np.array_equal(np.isnan(a), np.isnan(b)), i
        assert np.array_equal(np.nan_to_num(a, nan=-1.0), np.nan_to_num(b, nan=-1.0)), i
    for a, b in zip(wi_h, wi_d):
        assert np.array_equal(np.nan_to_num(a, nan=-1.0), np.nan_to_num(b, nan=-1.0))
    assert np.isfinite(dev[4]).all()                                          # NaN -> 0
    # video 1: +-inf -> the dtype's max / min; 65504 is harmless, 3.4e38 overflows inside the first projection (in the reference too)
    assert np.isfinite(dev[1]).all() if dtype == np.float16 else np.isnan(dev[1][256:]).all()
    assert np.isnan(dev[3]).all()                                             # inf without NaN: left alone, poisons its chunk
    assert all(np.isfinite(dev[i]).all() for i in (0, 2, 5))
    # nan_to_num=False: rows are used as they are
    raw = ragged(model, vids, nan_to_num=False)
    off = np.concatenate([[0], np.cumsum(lengths)])
    lg = raw["logits"].cpu().numpy()
    assert np.isnan(lg[off[1]:off[2]]).all() and np.isfinite(lg[off[0]:off[1]]).all()


@pytest.mark.parametrize("dtype", [np.float16, np.float32])
@pytest.mark.parametrize("micro_batch", [2, 3])
def test_nan_rule_is_decided_per_video_across_micro_batch_passes(dtype, micro_batch):
    """test.py:90-95 looks at the WHOLE video tensor.  Video 1 (6 chunks) holds -inf in chunk 0 and +inf in chunk 1 and its only
    NaN in chunk 3; with micro_batch = 2 / 3 those chunks are in different passes of the call, so the flag must be known before
    the first pass lays its rows out.  The reference replaces the infs (fp16: +-65504 -> finite scores everywhere; fp32: 3.4e38
    overflows inside in_proj, NaN chunk in the reference too).  Video 2 has an inf in its event features and no NaN: left alone.
    Compared with the host rule (`harness._unpack_item`: torch.isnan(x).any() / torch.nan_to_num on the padded tensor) followed
    by the dense forward -- bit for bit including the NaN pattern."""
    lengths = [40, 1400, 300, 256]
    vids = videos(lengths, seed=21, dtype=dtype)
    vids[1][0][3, 9] = -np.inf          # chunk 0
    vids[1][0][300, 100] = np.inf       # chunk 1
    vids[1][0][800, 5] = np.nan         # chunk 3: a later pass
    vids[2][1][10, 10] = np.inf         # no NaN in this video: stays inf
    model, _ = make_model("f32", outputs="scores", micro_batch=micro_batch)
    fixed = []
    for i, (img, ev) in enumerate(vids):
        ci, n = harness.process_split(img, 256)
        ce, _ = harness.process_split(ev, 256)
        item = (torch.tensor(ci).unsqueeze(0), torch.tensor(ce).unsqueeze(0), ("Normal",), torch.tensor([n]))
        a, b, _, _ = harness._unpack_item(item, 256, "ucfcrime", None)
        fixed.append((a.reshape(-1, 768)[:n].numpy().copy(), b.reshape(-1, 768)[:n].numpy().copy()))
    assert np.isfinite(fixed[1][0].astype(np.float64)).all() and np.isinf(fixed[2][1]).any()
    want = dense_reference(model, fixed)
    got = ragged(model, vids)
    for k in want:
        a, b = want[k].cpu().numpy(), got[k].cpu().numpy()
        assert np.array_equal(np.isnan(a), np.isnan(b)), k
        assert np.array_equal(np.nan_to_num(a, nan=-1.0), np.nan_to_num(b, nan=-1.0)), k
    off = np.concatenate([[0], np.cumsum(lengths)])
    lg = got["logits"].cpu().numpy()
    if dtype == np.float16:
        assert np.isfinite(lg[off[1]:off[2]]).all()                # the reference's fp16 answer: +-65504, finite scores
    else:
        assert np.isnan(lg[off[1]:off[1] + 512]).all() and np.isfinite(lg[off[1] + 512:off[2]]).all()
    assert np.isnan(lg[off[2]:off[2] + 256]).all() and np.isfinite(lg[off[2] + 256:off[3]]).all()
    assert np.isfinite(lg[off[0]:off[1]]).all() and np.isfinite(lg[off[3]:]).all()


def test_forward_videos_matches_the_oracle_and_rejects_bad_arguments():
    model, sd = make_model("f32", outputs="scores")
    vids = videos([90, 257, 30], seed=10)
    got = ragged(model, vids)
    torch.set_num_threads(harness.host_cpu_share())
    want = np.concatenate(orc.score_videos(orc.OracleMMFMIL(sd, orc.OracleConfig(num_refinement_steps=3)), vids))
    assert np.abs(torch.sigmoid(got["logits"]).cpu().numpy() - want).max() <= H.TOL_SIGMOID
    img = torch.zeros(10, 768, device="cuda")
    with pytest.raises(ValueError, match="sum of lengths"):
        model.forward_videos(img, img, [4, 5])
    with pytest.raises(ValueError, match="at least one snippet"):
        model.forward_videos(img, img, [10, 0])
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        model.forward_videos(img.cpu(), img.cpu(), [10])


@pytest.mark.parametrize("compute,batch_chunks", [("f32", 4), ("f32", 128), ("bf16", 8)])
def test_host_list_entry_equals_forward_videos(compute, batch_chunks):
    """`iefvad_forward_videos_host` (csrc/hostpipe.h): the library walks the list -- packs whole videos into passes, stages pass k + 1
    on a worker thread while pass k is sent and computed -- from the HOST tensors a DataLoader delivers (zero-padded chunk tensors
    whose first `len` rows count).  Bit-identical to one `forward_videos` call over the same rows whatever the pass sizes (the
    library ramps them up and tapers them): the f32 mode is bit-reproducible across batch sizes, and the bf16 mode's ring and
    row-block kernels are bit-identical to each other (only the fusion weights' row sums are added in another order).  Called
    twice: the second call reuses the handle's staging slots and copy threads."""
    lengths = EDGE_LENGTHS + [90, 400, 33]
    vids = videos(lengths, seed=23)
    model, _ = make_model(compute, outputs="scores")
    padded_i = [torch.from_numpy(harness.process_split(v[0], 256)[0]) for v in vids]
    padded_e = [torch.from_numpy(harness.process_split(v[1], 256)[0]) for v in vids]
    for _ in range(2):
        got = model.forward_videos_host(padded_i, padded_e, lengths, batch_chunks=batch_chunks)
    want = ragged(model, vids)
    for k in want:
        assert got[k].shape == (sum(lengths),)
        if compute == "bf16" and k != "logits":
            assert float((got[k] - want[k]).abs().max()) <= 1e-6, k          # row sums of the fusion weights: order depends on the kernel
        else:
            assert torch.equal(got[k], want[k]), (k, (got[k] - want[k]).abs().max().item())
    with pytest.raises(ValueError, match="contiguous host tensors"):
        model.forward_videos_host([padded_i[0].cuda()], [padded_e[0]], [lengths[0]])
    with pytest.raises(ValueError, match="same videos"):
        model.forward_videos_host(padded_i, padded_e[:-1], lengths)


def test_host_list_copy_pool_grows_between_calls():
    """ADVICE round 4 (csrc/hostgather.h): one handle, `host_threads` 2, then 8, then 16, on a list whose passes are large enough
    (> 1 MB of rows per modality) for the pool to split them -- threads added by the later calls must start at the pool's current
    generation, not re-run the earlier call's (freed) tables.  Every call's vectors equal `forward_videos` bit for bit."""
    lengths = [700, 1500, 300, 1024, 90, 2000, 513, 256, 1200, 800]
    vids = videos(lengths, seed=31)
    model, _ = make_model("f32", outputs="scores")
    padded_i = [torch.from_numpy(harness.process_split(v[0], 256)[0]) for v in vids]
    padded_e = [torch.from_numpy(harness.process_split(v[1], 256)[0]) for v in vids]
    want = ragged(model, vids)
    for threads in (2, 8, 16, 3):
        got = model.forward_videos_host(padded_i, padded_e, lengths, batch_chunks=8, host_threads=threads)
        for k in want:
            assert torch.equal(got[k], want[k]), (threads, k)


@pytest.mark.parametrize("compute", ["f32", "bf16"])
def test_host_list_passes_closed_at_one_round_of_row_blocks(compute):
    """A list long enough (31 k rows, 140 chunks) that the walk closes passes by ROWS: a row-compressed pass stops before its row set
    outgrows one 64-row block per CU (csrc/hostpipe.h), so the cuts fall at other videos than the chunk count alone would put them.
    Same results as one `forward_videos` call over all rows (bit for bit: f32 is batch-size invariant, the bf16 row-block and ring
    kernels agree with each other)."""
    rng = np.random.default_rng(5)
    lengths = [int(n) for n in rng.integers(300, 1100, 45)]
    vids = videos(lengths, seed=31)
    model, _ = make_model(compute, outputs="scores")
    padded_i = [torch.from_numpy(harness.process_split(v[0], 256)[0]) for v in vids]
    padded_e = [torch.from_numpy(harness.process_split(v[1], 256)[0]) for v in vids]
    got = model.forward_videos_host(padded_i, padded_e, lengths, batch_chunks=128)
    want = ragged(model, vids)
    for k in want:
        if compute == "bf16" and k != "logits":
            assert float((got[k] - want[k]).abs().max()) <= 1e-6, k
        else:
            assert torch.equal(got[k], want[k]), (k, (got[k] - want[k]).abs().max().item())


def test_host_list_bf16_wire_equals_forward_videos_on_rounded_rows():
    """`wire_dtype = BF16` (include/iefvad.h; SURVEY 7-2's down-conversion on the wire in the throughput mode): the staging threads
    round the fp32 rows to bf16 -- nearest even, NaN kept -- so half the bytes cross PCIe.  Bit for bit what `forward_videos` gives on
    rows torch rounded to bf16 (the same rounding; incl. a video with a NaN under the NaN rule, an -inf, values that round up to the
    next binade and to +-inf), for pass sizes that split the list at different videos; within the bf16 mode's own gate of the
    fp32-wire scores; refused for another compute mode or another dtype pair."""
    lengths = EDGE_LENGTHS + [90, 400, 33, 1300]
    vids = videos(lengths, seed=29)
    vids[2][0][3, 5] = np.nan                      # NaN rule: this video's image tensor -> nan_to_num (NaN -> 0, -inf -> lowest)
    vids[2][0][7, 9] = -np.inf
    vids[2][0][8, 1] = 3.4e38                      # finite in fp32, +inf once rounded to bf16: replaced under the same rule
    vids[4][1][0, :4] = [np.float32(1.0) + np.float32(2.0 ** -8), 1.0e-39, -65504.0, np.float32(1.0) - np.float32(2.0 ** -9)]
    model, _ = make_model("bf16", outputs="scores")
    padded_i = [torch.from_numpy(harness.process_split(v[0], 256)[0]) for v in vids]
    padded_e = [torch.from_numpy(harness.process_split(v[1], 256)[0]) for v in vids]
    img_b = torch.from_numpy(np.concatenate([v[0] for v in vids])).to(torch.bfloat16).cuda()
    ev_b = torch.from_numpy(np.concatenate([v[1] for v in vids])).to(torch.bfloat16).cuda()
    with torch.no_grad():
        want = model.forward_videos(img_b, ev_b, lengths)
    for batch_chunks in (3, 8, 128):
        got = model.forward_videos_host(padded_i, padded_e, lengths, batch_chunks=batch_chunks, wire_dtype=torch.bfloat16)
        for k in want:
            both_nan = torch.isnan(got[k]) & torch.isnan(want[k])      # the replaced +-inf (the bf16 extremes) overflow that chunk's LayerNorm
            if k == "logits":
                assert bool(((got[k] == want[k]) | both_nan).all()), (batch_chunks, k, (got[k] - want[k]).abs().nan_to_num().max().item())
            else:
                assert float((got[k] - want[k]).abs().nan_to_num().max()) <= 1e-6 and bool((torch.isnan(got[k]) == torch.isnan(want[k])).all()), (batch_chunks, k)
    assert int(torch.isnan(got["logits"]).sum()) == 256          # video 2 (one chunk of 256 rows), nothing else
    # against the fp32 wire: the bf16 mode's own accuracy class (the first layer's residual now reads the rounded row)
    clean = [i for i in range(len(lengths)) if i not in (2, 4)]
    ref = model.forward_videos_host([padded_i[i] for i in clean], [padded_e[i] for i in clean], [lengths[i] for i in clean])
    nar = model.forward_videos_host([padded_i[i] for i in clean], [padded_e[i] for i in clean], [lengths[i] for i in clean],
                                    wire_dtype=torch.bfloat16)
    d = (torch.sigmoid(ref["logits"]) - torch.sigmoid(nar["logits"])).abs()
    assert float(d.max()) <= 2e-2 and float(d.mean()) <= 2e-3, (float(d.max()), float(d.mean()))
    with pytest.raises(RuntimeError, match="wire_dtype"):
        model.forward_videos_host(padded_i, padded_e, lengths, wire_dtype=torch.float16)
    f32_model, _ = make_model("f32", outputs="scores")
    with pytest.raises(RuntimeError, match="bf16 mode"):
        f32_model.forward_videos_host(padded_i, padded_e, lengths, wire_dtype=torch.bfloat16)


@pytest.mark.parametrize("dtype", [np.float32, np.float16])
def test_score_loader_host_list_equals_the_python_loop(dtype):
    """harness.score_loader: the list walk inside the library (host_list=True, the default) against the Python loop around
    forward_videos (host_list=False) -- same batches, same scores bit for bit, incl. the NaN rule, an fp16 video inside an fp32 list
    (a dtype change closes a library call) and several calls per list (host_list_bytes)."""
    lengths = [100, 300, 50, 80, 600, 256, 17, 900]
    vids = videos(lengths, seed=29, dtype=dtype)
    vids[1][0][7, 5] = np.nan
    vids[1][0][290, 100] = np.inf
    vids[3][0][10, 10] = np.inf
    vids[4][1][400, 767] = np.nan
    if dtype == np.float32:
        vids[6] = (vids[6][0].astype(np.float16), vids[6][1].astype(np.float16))
    model, _ = make_model("f32", outputs="scores")

    def items():
        for img, ev in vids:
            ci, n = harness.process_split(img, 256)
            ce, _ = harness.process_split(ev, 256)
            yield torch.tensor(ci).unsqueeze(0), torch.tensor(ce).unsqueeze(0), ("Normal",), torch.tensor([n])

    loop, _, wi_l, we_l = harness.score_loader(model, items(), 256, "cuda:0", "ucfcrime", batch_chunks=4, host_list=False)
    for cap in (1 << 30, 1 << 20):
        lst, _, wi_h, we_h = harness.score_loader(model, items(), 256, "cuda:0", "ucfcrime", batch_chunks=4, host_list_bytes=cap)
        for i, (a, b) in enumerate(zip(loop, lst)):
            assert a.shape == (lengths[i],) and np.array_equal(np.isnan(a), np.isnan(b)), i
            assert np.array_equal(np.nan_to_num(a, nan=-1.0), np.nan_to_num(b, nan=-1.0)), i
        for a, b in zip(wi_l + we_l, wi_h + we_h):
            assert np.array_equal(np.nan_to_num(a, nan=-1.0), np.nan_to_num(b, nan=-1.0))
