"""The fp32-accurate split mode (compute="bf16x6", csrc/gemm_split.h): every dense projection is six bf16 MFMA
products of the exact three-term bf16 split of both fp32 operands.  It is held to the SAME gates as the fp32 MFMA
mode (tests/helpers.py TOL_*), and its error against an fp64 evaluation must not exceed the fp32 mode's.
Needs a real MI355X: run with `-m gpu`."""
import argparse
import ctypes as C

import numpy as np
import pytest
import torch

import iefvad_amd
from iefvad_amd import harness, synth
from oracle import iefvad_oracle as orc
from tests import helpers as H

pytestmark = pytest.mark.gpu

B_SPLIT = 48     # chunks: 12,288 rows -> every projection's 128 x 256 grid has >= 256 workgroups (the split kernel runs)


def make_model(sd, compute, L=2, K=10, nu=8, **kw):
    args = argparse.Namespace(visual_layers=L, visual_head=8, num_refinement_steps=K, lambda_ref=0.5,
                              noise_model="StudentT", nu=nu)
    m = iefvad_amd.MMFMIL(14, 768, 256, 768, 8, L, 8, 10, 10, "cuda", args, compute=compute, **kw)
    m.load_state_dict(sd)
    return m.to("cuda:0").eval()


def run(model, img, ev):
    with torch.no_grad():
        out = model(torch.from_numpy(img).cuda(), torch.from_numpy(ev).cuda(), None, None, None)
    torch.cuda.synchronize()
    return {k: v.cpu().numpy() for k, v in out.items()}


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def test_three_term_split_is_exact():
    """x == p0 + p1 + p2 for every finite fp32 x (8 + 9 + 9 significant bits); planes are bf16, round-to-nearest."""
    lib = iefvad_amd.lib.load_library()
    rng = np.random.default_rng(0)
    n = 1 << 20
    x = (rng.standard_normal(n) * np.exp(rng.uniform(-30, 30, n))).astype(np.float32)
    x[:8] = [0.0, -0.0, 1.0, -1.0, 3.0e38, -3.0e38, 1.1754944e-38, 1e-30]
    dx = torch.from_numpy(x).cuda()
    planes = torch.empty(3, n, dtype=torch.bfloat16, device="cuda")
    assert lib.iefvad_split_bf16x3(dx.data_ptr(), planes.data_ptr(), n, _stream()) == 0, iefvad_amd.lib.last_error()
    torch.cuda.synchronize()
    p = planes.to(torch.float64).cpu().numpy()
    assert np.array_equal(p[0] + p[1] + p[2], x.astype(np.float64))
    assert np.array_equal(planes[0].float().cpu().numpy(), torch.from_numpy(x).to(torch.bfloat16).float().numpy())   # RNE
    nz = x != 0
    assert (np.abs(p[1][nz]) <= np.abs(x[nz]).astype(np.float64) * 2.0 ** -8).all()        # |x - bf16(x)| <= half an ulp
    assert (np.abs(p[2][nz]) <= np.abs(x[nz]).astype(np.float64) * 2.0 ** -16).all()


def test_many_matrix_split_and_transposed_split_equal_the_single_split():
    """iefvad_split_bf16x3_many (one launch for every projection matrix of a model and, with rows > 0, for their transposes -- what
    iefvad_set_weights and the training backward call): bit for bit the planes iefvad_split_bf16x3 gives for the matrix, respectively
    for its torch transpose; 34 entries = two launches."""
    lib = iefvad_amd.lib.load_library()
    gen = torch.Generator(device="cuda").manual_seed(5)
    shapes = [(2304, 768), (768, 768), (1536, 768)] * 5 + [(768, 768)] * 2
    mats = [torch.randn(r, c, device="cuda", generator=gen) * float(np.exp(i - 8)) for i, (r, c) in enumerate(shapes)]
    srcs, dsts, ns, rows, want = [], [], [], [], []
    for i, m in enumerate(mats):
        for transposed in (False, True):
            out = torch.empty(3, m.numel(), dtype=torch.bfloat16, device="cuda")
            ref = torch.empty_like(out)
            base = m.t().contiguous() if transposed else m
            assert lib.iefvad_split_bf16x3(base.data_ptr(), ref.data_ptr(), m.numel(), _stream()) == 0, iefvad_amd.lib.last_error()
            srcs.append(m.data_ptr()); dsts.append(out.data_ptr()); ns.append(m.numel()); rows.append(m.shape[0] if transposed else 0)
            want.append((ref, out))
    cnt = len(srcs)
    a_src = (C.c_void_p * cnt)(*srcs); a_dst = (C.c_void_p * cnt)(*dsts)
    a_n = (C.c_size_t * cnt)(*ns); a_rows = (C.c_int32 * cnt)(*rows)
    assert lib.iefvad_split_bf16x3_many(a_src, a_dst, a_n, a_rows, cnt, _stream()) == 0, iefvad_amd.lib.last_error()
    torch.cuda.synchronize()
    for j, (ref, out) in enumerate(want):
        assert torch.equal(ref.view(torch.int16), out.view(torch.int16)), (j, rows[j])
    a_rows[1] = 100      # not a multiple of 64
    assert lib.iefvad_split_bf16x3_many(a_src, a_dst, a_n, a_rows, cnt, _stream()) != 0
    assert "transposed split" in iefvad_amd.lib.last_error()


def test_split_gemm_is_at_least_as_accurate_as_fp32_mfma():
    """C = A W^T + b: error of the split kernel vs an fp64 product <= error of the fp32 MFMA kernel (x 1.25 slack) on
    the three projection shapes, with asymmetric operands (a transposed accumulator map would show)."""
    lib = iefvad_amd.lib.load_library()
    rng = np.random.default_rng(1)
    for (M, N, K) in [(128, 256, 64), (1024, 768, 768), (512, 2304, 768), (256, 1536, 768)]:
        A = (rng.standard_normal((M, K)) * 1.3).astype(np.float32)
        W = (rng.uniform(-1, 1, (N, K)) / np.sqrt(K)).astype(np.float32)
        b = rng.standard_normal(N).astype(np.float32)
        dA, dW, db = (torch.from_numpy(x).cuda() for x in (A, W, b))
        planes = torch.empty(3, N, K, dtype=torch.bfloat16, device="cuda")
        assert lib.iefvad_split_bf16x3(dW.data_ptr(), planes.data_ptr(), N * K, _stream()) == 0
        Cs = torch.full((M, N), float("nan"), device="cuda")
        Cf = torch.full((M, N), float("nan"), device="cuda")
        assert lib.iefvad_gemm_bias(dA.data_ptr(), planes.data_ptr(), db.data_ptr(), Cs.data_ptr(), M, N, K, 2, _stream()) == 0, \
            iefvad_amd.lib.last_error()
        assert lib.iefvad_gemm_bias(dA.data_ptr(), dW.data_ptr(), db.data_ptr(), Cf.data_ptr(), M, N, K, 0, _stream()) == 0
        torch.cuda.synchronize()
        ref = A.astype(np.float64) @ W.astype(np.float64).T + b
        es = np.abs(Cs.cpu().numpy() - ref)
        ef = np.abs(Cf.cpu().numpy() - ref)
        assert es.max() <= 1.25 * ef.max() + 1e-7, (M, N, K, es.max(), ef.max())
        assert np.sqrt((es ** 2).mean()) <= 1.1 * np.sqrt((ef ** 2).mean()) + 1e-9, (M, N, K)


def test_forward_meets_the_fp32_gates_and_the_fp32_modes_error():
    """B = 48 chunks (the split kernel takes every projection): all eight outputs against the oracle within the fp32
    gates; against the fp64 oracle the split mode's error is not larger than the fp32 MFMA mode's (x 1.25)."""
    sd = synth.make_state_dict(0)
    img, ev = synth.make_inputs(7, B_SPLIT)
    cfg = orc.OracleConfig(num_layers=2, num_refinement_steps=10, nu=8)
    ti, te = torch.from_numpy(img), torch.from_numpy(ev)
    ref32 = orc.forward(sd, ti, te, cfg)
    ref64 = orc.forward(sd, ti, te, cfg, dtype=torch.float64)
    got = run(make_model(sd, "bf16x6"), img, ev)
    f32 = run(make_model(sd, "f32"), img, ev)
    for k in H.BIG_KEYS + ["logits"]:
        r32 = ref32[k].numpy()
        assert np.abs(got[k] - r32).max() <= (H.TOL_LOGIT if k == "logits" else H.TOL_BIG), k
        r64 = ref64[k].numpy()
        e_split, e_f32 = np.abs(got[k] - r64).max(), np.abs(f32[k] - r64).max()
        assert e_split <= 1.25 * e_f32 + 2e-7, (k, e_split, e_f32)
    assert np.abs(H.sigmoid(got["logits"]) - H.sigmoid(ref32["logits"].numpy())).max() <= H.TOL_SIGMOID
    assert not np.array_equal(got["logits"], f32["logits"])          # the split kernel really ran


def test_scores_only_outputs_equal_full_outputs():
    sd = synth.make_state_dict(3)
    img, ev = synth.make_inputs(11, B_SPLIT)
    full = run(make_model(sd, "bf16x6"), img, ev)
    sc = run(make_model(sd, "bf16x6", outputs="scores"), img, ev)
    assert np.array_equal(full["logits"], sc["logits"])
    assert np.allclose(full["w_i"].mean(-1), sc["w_i_mean"], atol=1e-6)


def test_small_batches_run_on_the_fp32_kernels():
    """Documented dispatch: a batch below 6 chunks uses the fp32 MFMA kernels (bit-identical to compute="f32"); K=5 /
    Gaussian variant and fp16 inputs go through the split path at B = 48."""
    sd = synth.make_state_dict(1)
    img, ev = synth.make_inputs(5, 3)
    a = run(make_model(sd, "bf16x6"), img, ev)
    b = run(make_model(sd, "f32"), img, ev)
    for k in a:
        assert np.array_equal(a[k], b[k]), k
    sd5 = synth.make_state_dict(2, 768, 2, 5)
    img, ev = synth.make_inputs(9, B_SPLIT)
    img16, ev16 = img.astype(np.float16), ev.astype(np.float16)
    cfg = orc.OracleConfig(num_layers=2, num_refinement_steps=5, nu=8)
    ref = orc.forward(sd5, torch.from_numpy(img16).float(), torch.from_numpy(ev16).float(), cfg)
    got = run(make_model(sd5, "bf16x6", K=5), img16, ev16)
    for k in H.BIG_KEYS:
        assert np.abs(got[k] - ref[k].numpy()).max() <= H.TOL_BIG, k
    assert np.abs(H.sigmoid(got["logits"]) - H.sigmoid(ref["logits"].numpy())).max() <= H.TOL_SIGMOID


@pytest.mark.parametrize("B", [5, 6, 9, 20])
def test_mid_size_batches_meet_the_fp32_gates_on_either_side_of_the_dispatch_rule(B):
    """The split kernels take a micro-batch from 6 chunks on (72 workgroups of a 768-wide projection: the measured crossover,
    tools/split_threshold_probe.py); B = 5 still runs on the fp32 kernels (bit-identical to compute="f32"), B = 6, 9, 20 on
    partially filled split grids.  All against the CPU oracle at the fp32 gates."""
    sd = synth.make_state_dict(4)
    img, ev = synth.make_inputs(21, B)
    got = run(make_model(sd, "bf16x6"), img, ev)
    f32 = run(make_model(sd, "f32"), img, ev)
    if B < 6:
        for k in got:
            assert np.array_equal(got[k], f32[k]), k
    else:
        assert not np.array_equal(got["logits"], f32["logits"])          # another arithmetic ran
    ref = orc.forward(sd, torch.from_numpy(img), torch.from_numpy(ev), orc.OracleConfig(num_layers=2, num_refinement_steps=10, nu=8))
    for k in H.BIG_KEYS:
        assert np.abs(got[k] - ref[k].numpy()).max() <= H.TOL_BIG, k
    assert np.abs(got["logits"] - ref["logits"].numpy()).max() <= H.TOL_LOGIT
    assert np.abs(H.sigmoid(got["logits"]) - H.sigmoid(ref["logits"].numpy())).max() <= H.TOL_SIGMOID


def test_dataset_scores_and_auc_match_the_oracle():
    """A UCF-shaped list scored with cross-video batching (64-chunk launches -> split kernel) against the fp32 oracle
    run per video: per-snippet |d score| <= 2e-6 and AUC / AP / Ano-AUC equal to 1e-6."""
    seed = 4
    lengths = synth.lognormal_lengths(seed, 60, 14000)
    classes = [synth.UCF_CLASSES[i % len(synth.UCF_CLASSES)] for i in range(len(lengths))]
    total = int(lengths.sum())
    gt = synth.make_gt(seed, total)
    sd = synth.make_state_dict(0)

    def items():
        for i, (n, c) in enumerate(zip(lengths, classes)):
            img, ev = synth.make_video(seed, i, int(n))
            ci, _ = harness.process_split(img, 256)
            ce, _ = harness.process_split(ev, 256)
            yield torch.tensor(ci).unsqueeze(0), torch.tensor(ce).unsqueeze(0), (c,), torch.tensor([int(n)])

    model = make_model(sd, "bf16x6", outputs="scores")
    s_gpu, c_gpu, _, _ = harness.score_loader(model, items(), 256, "cuda:0", "ucfcrime", batch_chunks=64)
    torch.set_num_threads(harness.host_cpu_share())
    oracle = orc.OracleMMFMIL(sd, orc.OracleConfig())
    s_cpu, c_cpu, _, _ = harness.score_loader(oracle, items(), 256, "cpu", "ucfcrime")
    assert c_gpu == c_cpu == classes
    a, b = np.concatenate(s_gpu), np.concatenate(s_cpu)
    assert a.shape == b.shape == (total,)
    assert np.abs(a - b).max() <= H.TOL_SIGMOID
    r_gpu = harness.evaluate_scores(s_gpu, classes, gt, "ucfcrime", verbose=False)
    r_cpu = harness.evaluate_scores(s_cpu, classes, gt, "ucfcrime", verbose=False)
    for k in ("roc", "ap", "ano_auc"):
        assert abs(r_gpu[k] - r_cpu[k]) <= 1e-6, (k, r_gpu[k], r_cpu[k])


def test_full_config4_batch_properties_bf16x6():
    """BASELINE config 4 at its full single-GPU size (B = 8192 chunks = 2,097,152 snippets resident in HBM) in the
    bench's default arithmetic, through size-independent properties: every score finite and in (0, 1); n_i + n_e = 1;
    a duplicated chunk in another micro-batch gives bit-identical scores (same kernels, row-independent arithmetic); the
    scores of sampled chunks agree with the fp32 MFMA mode within the fp32 gate; permuting whole micro-batches permutes
    the scores bit for bit."""
    sd = synth.make_state_dict(9)
    model = make_model(sd, "bf16x6", outputs="scores")
    g = torch.Generator(device="cuda:0")
    g.manual_seed(1234)
    B = 8192
    img = torch.randn(B, 256, 768, device="cuda:0", generator=g) * 0.45
    ev = torch.randn(B, 256, 768, device="cuda:0", generator=g) * 0.45
    img[4097], ev[4097] = img[11], ev[11]
    with torch.no_grad():
        out = model(img, ev, None, None, None)
    lg = out["logits"]
    assert lg.shape == (B, 256, 1) and bool(torch.isfinite(lg).all())
    assert torch.equal(lg[11], lg[4097])
    s = torch.sigmoid(lg)
    assert 0.0 < float(s.min()) and float(s.max()) < 1.0
    assert float((out["w_i_mean"] + out["w_e_mean"] - 1.0).abs().max()) < 1e-5
    # the same 64 chunks through the fp32 MFMA mode
    pick = torch.arange(0, B, 128, device="cuda:0")
    f32 = make_model(sd, "f32", outputs="scores")
    with torch.no_grad():
        ref = f32(img[pick].contiguous(), ev[pick].contiguous(), None, None, None)
    assert float((torch.sigmoid(ref["logits"]) - s[pick]).abs().max()) <= H.TOL_SIGMOID
    assert float((ref["logits"] - lg[pick]).abs().max()) <= H.TOL_LOGIT
    # ... and eight of them through the CPU oracle (the fp32 restatement of the reference pinned by tests/golden), so that the
    # bench-sized batch is checked against the reference's arithmetic and not only against another HIP mode
    pick8 = pick[::8]
    torch.set_num_threads(harness.host_cpu_share())
    o = orc.forward(sd, img[pick8].cpu(), ev[pick8].cpu(), orc.OracleConfig())
    assert float((torch.sigmoid(o["logits"]) - s[pick8].cpu()).abs().max()) <= H.TOL_SIGMOID
    assert float((o["logits"] - lg[pick8].cpu()).abs().max()) <= H.TOL_LOGIT
    assert float((o["w_i"].mean(-1) - out["w_i_mean"][pick8].cpu()).abs().max()) <= 1e-5
    # swap the first two 256-chunk micro-batches
    perm = torch.cat([torch.arange(256, 512), torch.arange(0, 256), torch.arange(512, 1024)]).to("cuda:0")
    with torch.no_grad():
        a = model(img[:1024].contiguous(), ev[:1024].contiguous(), None, None, None)
        b = model(img[:1024][perm].contiguous(), ev[:1024][perm].contiguous(), None, None, None)
    assert torch.equal(a["logits"][perm], b["logits"])
    del img, ev, out, a, b
    torch.cuda.empty_cache()
