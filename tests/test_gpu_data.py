"""On-device CIFAR-10 input pipeline vs the numpy restatement of the reference's transform chain (oracle/np_oracle.py:cifar_augment,
wqaq/dorefa/main.py:203-210): bit-exact for the same random draws; loader semantics (every sample once per epoch, reproducible from the seed);
checkpoint layout of the reference scripts."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_augment_bit_exact_vs_oracle():
    from micronet_amd import data
    from oracle import np_oracle as O
    r = np.random.default_rng(0)
    imgs = r.integers(0, 256, size=(64, 32, 32, 3), dtype=np.uint8)
    imgs[0] = 255; imgs[1] = 0
    B = 48
    idx = r.integers(0, 64, size=B).astype(np.int32)
    ox, oy = r.integers(0, 9, size=B).astype(np.int32), r.integers(0, 9, size=B).astype(np.int32)
    ox[:4], oy[:4] = [0, 8, 0, 8], [0, 0, 8, 8]                     # the four corner crops
    flip = (r.random(B) < 0.5).astype(np.uint8)
    got = data.augment(torch.from_numpy(imgs).cuda(), torch.from_numpy(idx).cuda(), torch.from_numpy(ox).cuda(), torch.from_numpy(oy).cuda(),
                       torch.from_numpy(flip).cuda()).cpu().numpy()
    ref = O.cifar_augment(imgs, idx, ox, oy, flip)
    assert np.array_equal(got, ref), float(np.abs(got - ref).max())


def test_loader_epoch_semantics_and_test_transform():
    from micronet_amd import data
    from oracle import np_oracle as O
    r = np.random.default_rng(1)
    n = 1000
    imgs = r.integers(0, 256, size=(n, 32, 32, 3), dtype=np.uint8)
    labels = np.arange(n) % 10
    ld = data.DeviceCifarLoader(imgs, labels, batch_size=256, train=True, seed=7)
    assert len(ld) == 4
    seen = []
    first = None
    for x, y in ld:
        assert x.shape[1:] == (3, 32, 32) and x.dtype == torch.float32 and y.dtype == torch.int64
        seen.append(y.cpu())
        first = x if first is None else first
    assert sorted(torch.cat(seen).tolist()) == sorted(labels.tolist())          # every sample exactly once
    ld2 = data.DeviceCifarLoader(imgs, labels, batch_size=256, train=True, seed=7)
    assert torch.equal(next(iter(ld2))[0], first)                                 # reproducible from the seed
    te = data.DeviceCifarLoader(imgs, labels, batch_size=500, train=False)
    xb, yb = next(iter(te))
    ref = O.cifar_augment(imgs, np.arange(500), np.full(500, 4), np.full(500, 4), np.zeros(500, dtype=np.uint8))
    assert np.array_equal(xb.cpu().numpy(), ref) and torch.equal(yb.cpu(), torch.from_numpy(labels[:500]))


def test_reference_checkpoint_layout_round_trip(tmp_path):
    from micronet_amd import data
    from micronet_amd.train import build_model
    from micronet.compression.quantization.wqaq.dorefa import quantize
    m = quantize.prepare(build_model("nin_gc"), inplace=True, a_bits=2, w_bits=2).cuda()
    dp = torch.nn.DataParallel(m)                    # the reference wraps the model: its state_dict keys carry a "module." prefix
    data.save_state(dp, 91.03, str(tmp_path / "nin_gc.pth"))
    st = torch.load(str(tmp_path / "nin_gc.pth"), map_location="cpu")
    assert set(st.keys()) == {"best_acc", "state_dict"} and not any(k.startswith("module.") for k in st["state_dict"])
    m2 = quantize.prepare(build_model("nin_gc", seed=3), inplace=True, a_bits=2, w_bits=2, fuse_blocks=False).cuda()
    assert data.load_state(m2, str(tmp_path / "nin_gc.pth")) == 91.03
    m.eval(), m2.eval()
    x = torch.randn(8, 3, 32, 32, device="cuda")
    with torch.no_grad():
        assert torch.allclose(m(x), m2(x), rtol=1e-6, atol=1e-7)
