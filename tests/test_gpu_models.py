"""Whole-net QAT steps on the MI355X vs the reference's CPU trajectory (tests/golden/models.npz, produced by the real
reference).  The nets are chaotic under quantisation (one flipped activation code changes downstream codes), so the
comparison is statistical: step-0 logits / loss / per-tensor gradient norms close, 3-step loss trajectory close."""
import importlib

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CFG = {
    "c1_nin_gc_dorefa_w8a8": ("nin_gc", "wqaq.dorefa", dict(a_bits=8, w_bits=8), 8, 1e-5),
    "c2_nin_gc_wbwtab_w3a2": ("nin_gc", "wbwtab", dict(A=2, W=3), 8, 0.0),
    "c2b_nin_gc_wbwtab_w2a2": ("nin_gc", "wbwtab", dict(A=2, W=2), 8, 0.0),
    "c3_nin_gc_iao_w8a8_bnfuse": ("nin_gc", "wqaq.iao", dict(a_bits=8, w_bits=8, q_type=0, q_level=0, weight_observer=0, bn_fuse=True), 8, 1e-5),
    "c4_resnet18_dorefa_w2a2": ("resnet18", "wqaq.dorefa", dict(a_bits=2, w_bits=2), 4, 1e-5),
    "c5_resnet18_iao_w4a4": ("resnet18", "wqaq.iao", dict(a_bits=4, w_bits=4, q_type=0, q_level=0), 4, 1e-5),
    "nin_dorefa_w4a4": ("nin", "wqaq.dorefa", dict(a_bits=4, w_bits=4), 4, 1e-5),
}


@pytest.mark.parametrize("key", list(CFG))
def test_training_trajectory_vs_reference(golden, key):
    from micronet_amd.train import build_model, make_optimizer, synth_batch, train_step
    arch, scheme, kw, B, wd = CFG[key]
    quantize = importlib.import_module("micronet.compression.quantization.%s.quantize" % scheme)
    model = quantize.prepare(build_model(arch), inplace=True, **kw).cuda()
    opt = make_optimizer(model, 0.01, wd)
    x, y = synth_batch(B, device="cuda")
    model.train()
    ref_losses = golden.meta["surface"][key]["losses"]
    losses = []
    for step in range(3):
        out = model(x)
        loss = torch.nn.functional.cross_entropy(out, y)
        opt.zero_grad()
        loss.backward()
        if step == 0:
            ref = golden.mo[f"{key}_logits0"]
            err = np.max(np.abs(out.detach().cpu().numpy() - ref)) / max(np.max(np.abs(ref)), 1e-6)
            assert err <= 2e-2, ("logits0", err)
            gn_ref = golden.meta["surface"][key]["gradnorm0"]
            bad = []
            for n_, p in model.named_parameters():
                r = gn_ref[n_]
                gnorm = float(p.grad.double().norm())
                if abs(gnorm - r) > 0.1 * max(r, 1e-6) + 1e-7:
                    bad.append((n_, gnorm, r))
            assert len(bad) <= max(1, len(gn_ref) // 20), bad[:5]
            mid = golden.meta["surface"][key]["mid"]
            gm = dict(model.named_parameters())[mid].grad[:8].cpu().numpy()
            rm = golden.mo[f"{key}_grad0_mid"]
            assert np.max(np.abs(gm - rm)) <= 0.1 * np.max(np.abs(rm)) + 1e-8
        opt.step()
        losses.append(float(loss))
    print(key, "gpu", losses, "ref", ref_losses)
    assert abs(losses[0] - ref_losses[0]) <= 2e-3
    assert all(abs(a - b) <= 0.15 * max(1.0, abs(b)) for a, b in zip(losses, ref_losses)), (losses, ref_losses)
    model.eval()
    out = model(x)
    assert torch.isfinite(out).all()


def test_state_dict_round_trip_with_reference_layout(golden):
    """state_dict produced on the GPU loads back and keeps the reference key layout (checkpoint interchange)."""
    from micronet_amd.train import build_model
    from micronet.compression.quantization.wqaq.iao import quantize
    model = quantize.prepare(build_model("nin_gc"), inplace=True, a_bits=8, w_bits=8, bn_fuse=True).cuda().train()
    x = torch.randn(4, 3, 32, 32, device="cuda")
    model(x)
    sd = {k: v.cpu() for k, v in model.state_dict().items()}
    assert [[k, list(v.shape)] for k, v in sd.items()] == golden.meta["surface"]["c3_nin_gc_iao_w8a8_bnfuse"]["state"]
    model2 = quantize.prepare(build_model("nin_gc", seed=5), inplace=True, a_bits=8, w_bits=8, bn_fuse=True).cuda()
    model2.load_state_dict(sd)
    model.eval(), model2.eval()
    assert torch.equal(model(x), model2(x))
