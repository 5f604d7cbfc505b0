"""Host-side packing of the fp32 first stage (ccedit_amd/vae_f32.py), no GPU: the four parity windows that replace
conv3x3(nearest_upsample_2x(x)) (model.py:56-71) reproduce it exactly in fp64 arithmetic, in the kernel's K layout."""
import torch
import torch.nn.functional as F


def test_parity_windows_reproduce_upsample_conv():
    from ccedit_amd import vae_f32 as V
    g = torch.Generator().manual_seed(3)
    n, cin, cout, h, w = 2, 12, 5, 6, 7
    x = torch.randn(n, cin, h, w, generator=g, dtype=torch.float64)
    wt = torch.randn(cout, cin, 3, 3, generator=g, dtype=torch.float64)
    b = torch.randn(cout, generator=g, dtype=torch.float64)
    ref = F.conv2d(F.interpolate(x, scale_factor=2.0, mode="nearest"), wt, b, padding=1)
    packs = V.pack_f32_parities(wt.float().double(), b, "cpu")          # (weights representable in fp32: the merged taps round once)
    assert len(packs) == 4 and all(p.taps == 4 and p.n == cout and p.cpad == 16 and p.cin == 12 for p in packs)
    out = torch.zeros_like(ref)
    for p, pk in enumerate(packs):
        py, px = p >> 1, p & 1
        w2 = pk.w.double().view(cout, 4, pk.cpad)[:, :, :cin].reshape(cout, 2, 2, cin).permute(0, 3, 1, 2)     # tap = 2 dy + dx
        assert float(pk.w.view(cout, 4, pk.cpad)[:, :, cin:].abs().max()) == 0.0
        # window of source pixel (y, x) starts at (y - 1 + py, x - 1 + px): pad one row / column on the side the window leaves the frame
        xp = F.pad(x, (1 - px, px, 1 - py, py))
        out[:, :, py::2, px::2] = F.conv2d(xp, w2, pk.bias.double())
    wt32 = wt.float().double()
    ref32 = F.conv2d(F.interpolate(x, scale_factor=2.0, mode="nearest"), wt32, b.float().double(), padding=1)
    err = float((out - ref32).abs().max() / ref32.abs().max())
    assert err < 5e-7, err          # one fp32 rounding per merged tap


def test_pack_f32_layouts():
    from ccedit_amd import vae_f32 as V
    w3 = torch.arange(2 * 5 * 9, dtype=torch.float32).view(2, 5, 3, 3)
    p = V.pack_f32(w3, None, "cpu")
    assert (p.taps, p.n, p.cin, p.cpad) == (9, 2, 8, 16) and p.w.shape == (2, 9 * 16)
    assert torch.equal(p.w.view(2, 9, 16)[1, 4, :5], w3[1, :, 1, 1])          # tap = 3 ky + kx, channels contiguous, zero padding
    assert float(p.w.view(2, 9, 16)[:, :, 5:].abs().max()) == 0.0
    p1 = V.pack_f32(torch.ones(3, 20), torch.zeros(3), "cpu")
    assert (p1.taps, p1.cin, p1.cpad) == (1, 20, 32)
