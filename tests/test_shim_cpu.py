"""The MMFMIL shim's module protocol (no GPU): state_dict layout, default initialisation and the
attributes the reference's callers rely on."""
import argparse
import os

import numpy as np
import pytest
import torch

import iefvad_amd
from iefvad_amd import synth


def make(K=10, L=2, **kw):
    args = argparse.Namespace(visual_layers=L, visual_head=8, num_refinement_steps=K, lambda_ref=0.5,
                              noise_model="StudentT", nu=8)
    return iefvad_amd.MMFMIL(14, 768, 256, 768, 8, L, 8, 10, 10, "cuda", args, **kw)


@pytest.mark.parametrize("K", [10, 0])
def test_seeded_default_init_equals_reference(golden_dir, K):
    """Same parameter names, shapes, registration order AND default init as the reference: a seeded
    construction draws the same RNG stream, so every tensor matches the reference's bit for bit."""
    g = np.load(os.path.join(golden_dir, "init_checksums.npz"))
    torch.manual_seed(123)
    sd = make(K).state_dict()
    assert list(sd.keys()) == [str(k) for k in g[f"keys_k{K}"]]
    sums = np.array([float(v.double().sum()) for v in sd.values()])
    head = np.stack([np.resize(v.reshape(-1)[:4].double().numpy(), 4) for v in sd.values()])
    # fp64 sums depend on the reduction order (thread count): compare to 1e-12, the leading elements exactly
    assert np.allclose(sums, g[f"sums_k{K}"], rtol=1e-12, atol=1e-12)
    assert np.array_equal(head, g[f"head_k{K}"])


def test_state_dict_roundtrip_and_counts():
    m = make(10)
    sd = synth.make_state_dict(5)
    assert [k for k, _, _ in synth.state_dict_keys(2, 10)] == list(m.state_dict().keys())
    res = m.load_state_dict(sd)            # bare state_dict, as test.py:377-378 loads it
    assert not res.missing_keys and not res.unexpected_keys
    assert len(sd) == 78 and sum(p.numel() for p in m.parameters()) == 23_633_665
    for k, v in m.state_dict().items():
        assert torch.equal(v, sd[k])


def test_attributes_used_by_callers():
    m = make(3)
    assert m.temporal.nu == 8                       # ucf_train.py:94-95
    assert m.temporal.num_refinement_steps == 3 and m.temporal.lambda_ref == 0.5
    assert m.temporal.epsilon == 1e-8               # --epsilon never reaches the model (imf_vad.py:30-38)
    assert m.eval() is m and not m.training
    with pytest.raises(ValueError):
        make(1, outputs="everything")


def test_lanes_share_parameters_and_notice_weight_changes():
    """MMFMIL.lanes(n): shallow copies for concurrent forwards on several HIP streams.  They must hold the SAME Parameter
    objects (one set of weights), own no library handle until they run, be cached, and be invalidated together with the
    parent by load_state_dict / refresh_weights.  No GPU needed."""
    m = make(2)
    lanes = m.lanes(3)
    assert len(lanes) == 3 and lanes[0] is m and lanes[1] is not m and lanes[1] is not lanes[2]
    assert m.lanes(3)[1] is lanes[1] and m.lanes(2) == lanes[:2]                    # cached, prefix-stable
    assert m.lanes(1) == [m]
    for c in lanes[1:]:
        assert c.temporal is m.temporal and c._handle is None and c._workspace is None
        assert [id(p) for p in c.parameters()] == [id(p) for p in m.parameters()]
        assert c.outputs == m.outputs and c.compute == m.compute
    for c in lanes:
        c._weights_sig = ("stale",)
    m.load_state_dict(synth.make_state_dict(9, 768, 2, 2))
    assert all(c._weights_sig is None for c in lanes)
    for c in lanes:
        c._weights_sig = ("stale",)
    m.refresh_weights()
    assert all(c._weights_sig is None for c in lanes)
    # the state_dict does not grow lane entries
    assert len(m.state_dict()) == len(make(2).state_dict())
