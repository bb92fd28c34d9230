"""fp8 training dumps (NERF_AMD_BF16_F8, `nerf_amd.set_train_dumps("fp8")`): in bf16 mode the hidden activations the training forwards
dump and the deltas the dgrad chains dump -- the two operands of every weight-gradient product -- are stored as OCP e4m3 with one
power-of-two scale per sample and 16-feature K group (288 instead of 512 bytes per sample and layer in each of the four passes).  The
arithmetic is untouched (bf16 x bf16 -> fp32 in the forward, the chain and the products); only the products' operands are rounded to 4
significant bits, independently per element, so the error of a gradient entry averages out over the samples it sums.

Checked here: (i) the stored activations ARE the bf16 activations rounded to e4m3 under the stated scale rule (layout and scale
semantics, decoded on the host); (ii) parameter gradients of a training step against the same step in fp64, next to the bf16-dump
gradients; (iii) the forward output is bit-identical (the dump format cannot change it)."""
import math

import pytest
import torch
import torch.nn.functional as F

import weights as W
from oracle import nerf_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def A():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import nerf_amd
    from nerf_amd import addtional, mip_methods, mip_model, nerf_base, ops, utils

    class NS:
        pass
    ns = NS()
    ns.pkg, ns.ops, ns.addtional, ns.mip_methods, ns.mip_model, ns.nerf_base, ns.utils = nerf_amd, ops, addtional, mip_methods, mip_model, nerf_base, utils
    yield ns
    nerf_amd.set_train_dumps("bf16")
    nerf_amd.set_precision("fp32")


def _nets(A, tag):
    prop, mip = A.addtional.ProposalNetwork(10, 256), A.mip_model.MipNeRF(10, 4, 256)
    prop.load_state_dict(W.proposal_state(tag))
    mip.load_state_dict(W.mip_state(tag))
    return prop.cuda().train(), mip.cuda().train()


def _decode_f8_slot(dump, layer, layer_stride, n_sub, n_kg=16):
    """host-side reading of one fp8 slot -> (n_sub * 32, 16 * n_kg) float32 rows (mlp_layout.h: 8 data blocks + scale exponents per subtile;
    element e of K group kg in lane j + 32 h = feature 16 kg + 8 (e >> 2) + 4 h + (e & 3) of sample j)"""
    raw = dump[layer * layer_stride: layer * layer_stride + n_sub * 9216].view(n_sub, 9216)
    data = raw[:, :8192].reshape(n_sub, 8, 64, 2, 8)                       # [sub][fb][lane][kg & 1][e]
    vals = data.view(torch.float8_e4m3fn).float()
    ex = raw[:, 8192:].reshape(n_sub, 64, 16).float()                      # [sub][lane][kg]
    scale = torch.exp2(ex - 127.0).permute(0, 2, 1).reshape(n_sub, 8, 2, 64).permute(0, 1, 3, 2)      # -> [sub][fb][lane][kg & 1]
    vals = vals * scale[..., None]
    v = vals.permute(0, 1, 3, 2, 4).reshape(n_sub, 16, 2, 32, 2, 4)        # [sub][kg][h][j][e >> 2][e & 3]
    rows = v.permute(0, 3, 1, 4, 2, 5).reshape(n_sub * 32, 16 * 16)        # feature = 16 kg + 8 (e >> 2) + 4 h + (e & 3)
    return rows[:, : 16 * n_kg]


@pytest.mark.parametrize("tag", ["small", "he"])
def test_fp8_activation_dump_is_the_rounded_bf16_dump(A, tag):
    prop, mip = _nets(A, tag)
    ops = A.ops
    M = 1000
    g = torch.Generator().manual_seed(5)
    pts = torch.cat((torch.randn(M, 3, generator=g) * 1.5, torch.randn(M, 3, generator=g)), -1).cuda()
    pk = mip.packed(ops.BF16)
    out16, dump16 = ops.mip_forward_train(pk, ops.BF16, pts)
    out8, dump8 = ops.mip_forward_train(pk, ops.BF16_F8, pts)
    assert torch.equal(out16, out8) and dump8.numel() == dump16.numel()
    n_sub = (M + 255) // 256 * 8
    ls = n_sub * 16 * 1024
    for layer, width in ((0, 256), (3, 256), (6, 256), (7, 128)):
        want = ops.train_dump_rows(dump16, ops.NET_MIP, ops.BF16, M, layer, width).float()
        got = _decode_f8_slot(dump8, layer, ls, n_sub, width // 16)[:M].cuda()
        # per (sample, K group): scale 2^(e_max - 7) puts the largest magnitude into [128, 256) -> e4m3 spacing 16 there, i.e. every
        # element is within half a step of 2^(e_max - 3) ... relative to the group maximum: <= 2^-4; elements far below the maximum
        # lose relative precision (subnormal range), which is the format's contract
        grp = want.view(M, width // 16, 16).abs().amax(-1, keepdim=True)
        err = (got.view(M, width // 16, 16) - want.view(M, width // 16, 16)).abs()
        assert bool((err <= grp * (2.0 ** -4) + 1e-30).all()), (layer, float((err / (grp + 1e-30)).max()))
        big = want.view(M, width // 16, 16).abs() >= grp * 0.25                   # the normal range: relative error <= 2^-4
        rel = (err / want.view(M, width // 16, 16).abs().clamp_min(1e-30))[big]
        assert float(rel.max()) <= 2.0 ** -4 + 1e-6
    # the encoding slot and the mask bits are the same bytes in both formats
    enc = lambda d: d[8 * ls: 8 * ls + n_sub * 16 * 1024].view(n_sub, 16, 1024)[:, :6]      # (K groups 0..5 are written: PE10 + PE4)
    assert torch.equal(enc(dump8), enc(dump16))
    assert torch.equal(dump8[9 * ls: 9 * ls + 8 * n_sub * 1024], dump16[9 * ls: 9 * ls + 8 * n_sub * 1024])      # mask bits of slots 0..7
    # proposal network
    pkp = prop.packed(ops.BF16)
    d16, pd16 = ops.proposal_forward_train(pkp, ops.BF16, pts[:, :3].contiguous())
    d8, pd8 = ops.proposal_forward_train(pkp, ops.BF16_F8, pts[:, :3].contiguous())
    assert torch.equal(d16, d8)
    want = ops.train_dump_rows(pd16, ops.NET_PROPOSAL, ops.BF16, M, 2, 256).float()
    got = _decode_f8_slot(pd8, 2, ls, n_sub)[:M].cuda()
    grp = want.view(M, 16, 16).abs().amax(-1, keepdim=True)
    assert bool(((got - want).view(M, 16, 16).abs() <= grp * (2.0 ** -4) + 1e-30).all())


def _step(A, prop, mip, rays, z_c, tgt, u_inv, f_n):
    R, Zc = rays.cuda(), z_c.cuda()
    pts = (R[:, None, :3] + R[:, None, 3:] * Zc[:, :, None]).contiguous()
    dens = F.softplus(prop.forward(pts))
    pw = A.mip_methods.maxBlurFilter(A.addtional.ProposalNetwork.get_weights(dens, Zc, R[:, 3:]), 0.01)
    z_all, below = A.utils.inverseSample(pw, Zc, f_n + 1, sort=True, u=u_inv)
    z_f = z_all[..., :-1].contiguous()
    rgbo = mip.forward(A.nerf_base.NeRF.length2pts(R, z_f))
    rend, wts, _ = A.nerf_base.NeRF.render(rgbo, z_f, R[:, 3:])
    loss = torch.mean((rend - tgt.cuda()) ** 2) + A.addtional.ProposalLoss()(A.addtional.getBounds(pw, below), wts.detach())
    mip.zero_grad(); prop.zero_grad()
    loss.backward()
    grads = {"mip." + k: v.grad.detach().cpu().double() for k, v in mip.named_parameters()}
    grads.update({"prop." + k: v.grad.detach().cpu().double() for k, v in prop.named_parameters()})
    return loss.item(), z_f.detach().cpu(), below.cpu(), grads


@pytest.mark.parametrize("tag", ["small", "he"])
def test_fp8_dump_gradients_against_fp64(A, tag):
    """One training step (train.py:164-199) in bf16 mode with bf16 dumps and with fp8 dumps, same batch: every parameter gradient against
    the fp64 evaluation of the oracle's expressions on the same fine depths.  Measure: |g - g64|_2 / |g64|_2 per tensor (the quantity an
    optimizer step sees).  The fp8 operands may cost at most a small multiple of what bf16 arithmetic already costs."""
    prop, mip = _nets(A, tag)
    n, c_n, f_n = 1024, 32, 64
    g = torch.Generator().manual_seed(11)
    o = torch.tensor([0.0, 0.0, 4.0]).expand(n, 3)
    d = torch.randn(n, 3, generator=g) * 0.25 + torch.tensor([0.0, 0.0, -1.0])
    rays = torch.cat((o, d), -1).contiguous()
    tgt = torch.rand(n, 3, generator=g)
    res = 4.0 / c_n
    z_c = torch.linspace(2.0, 6.0 - res, c_n) + torch.rand(n, c_n, generator=g) * res
    u_inv = torch.rand(n, f_n + 1, generator=g)
    A.pkg.set_precision("bf16")
    A.pkg.set_train_dumps("bf16")
    l16, zf16, below16, g16 = _step(A, prop, mip, rays, z_c, tgt, u_inv, f_n)
    A.pkg.set_train_dumps("fp8")
    l8, zf8, below8, g8 = _step(A, prop, mip, rays, z_c, tgt, u_inv, f_n)
    A.pkg.set_train_dumps("bf16")
    assert l8 == l16 and torch.equal(zf8, zf16)                            # the forward does not depend on the dump format
    # the same step in fp64 (oracle expressions, torch.autograd), fine depths and bin indices from the run above
    cast = lambda sd: {k: v.double().requires_grad_(True) for k, v in sd.items()}
    p64, m64 = cast(W.proposal_state(tag)), cast(W.mip_state(tag))
    r, zc, zf = rays.double(), z_c.double(), zf16.double()
    dens = F.softplus(O.proposal_forward(p64, r[:, None, :3] + r[:, None, 3:] * zc[:, :, None]))
    pw = O.max_blur(O.sigma_to_weights(dens, zc, r[:, 3:]), 0.01)
    rend, wts, _ = O.composite(O.mip_forward(m64, O.length2pts(r, zf)), zf, r[:, 3:])
    (torch.mean((rend - tgt.double()) ** 2) + O.proposal_loss(O.get_bounds(pw, below16), wts.detach())).backward()
    exact = {"mip." + k: v.grad for k, v in m64.items()}
    exact.update({"prop." + k: v.grad for k, v in p64.items()})
    rep, worst = {}, 0.0
    for k in exact:
        nrm = exact[k].norm().item()
        e16, e8 = (g16[k] - exact[k]).norm().item() / nrm, (g8[k] - exact[k]).norm().item() / nrm
        cos = float((g8[k] * exact[k]).sum() / (g8[k].norm() * exact[k].norm()))
        cos16 = float((g16[k] * exact[k]).sum() / (g16[k].norm() * exact[k].norm()))
        rep[k] = (e16, e8, cos)
        worst = max(worst, e8)
        assert e8 <= max(3.0 * e16, 3e-2), (k, e16, e8)
        assert cos >= min(0.999, cos16 - 0.005), (k, cos, cos16)        # (first-layer tensors of the 'small' set nearly cancel: bf16 itself is at ~0.995)
    print("\n[%s] relative L2 error of the bf16-mode gradients vs fp64 (bf16 dumps | fp8 dumps | cosine of the fp8 one):" % tag)
    for k, (e16, e8, cos) in rep.items():
        print("   %-28s %.2e | %.2e | %.6f" % (k, e16, e8, cos))


def test_fp8_dumps_ragged_sizes_and_training_iteration(A):
    """ragged sample counts (partial subtiles) through forward + chain + products; and a TrainStep with fp8 dumps learns like the bf16 one."""
    ops = A.ops
    A.pkg.set_precision("bf16")
    prop, mip = _nets(A, "he")
    for M in (1, 33, 255, 257, 70001):
        g = torch.Generator().manual_seed(M)
        pts = torch.cat((torch.randn(M, 3, generator=g), torch.randn(M, 3, generator=g)), -1).cuda().requires_grad_(False)
        gout = torch.randn(M, 4, generator=g).cuda()
        res = {}
        for fmt in ("bf16", "fp8"):
            A.pkg.set_train_dumps(fmt)
            mip.zero_grad()
            (mip.forward(pts) * gout).sum().backward()
            res[fmt] = torch.cat([p.grad.reshape(-1) for p in mip.parameters()]).double()
        rel = (res["fp8"] - res["bf16"]).norm().item() / res["bf16"].norm().item()
        assert math.isfinite(rel) and rel <= (0.2 if M < 64 else 0.05), (M, rel)
    A.pkg.set_train_dumps("bf16")
