#!/usr/bin/env python3
"""Stage-by-stage diagnostic of the hand-written MLP backward (bwd_kernels.hip) against torch on the SAME activation dump:
delta of every chain layer, every weight / bias gradient.  Prints relative errors; exits non-zero above the gates."""
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
sys.path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))), "tests"))
import nerf_amd
import weights as W
from nerf_amd import ops
sys.path.insert(0, __import__('os').path.join(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))), 'tests'))
import torch_spec as ab                                 # the ops' torch specifications (test infrastructure)
from nerf_amd.addtional import ProposalNetwork
from nerf_amd.mip_model import MipNeRF


def rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


def main():
    bad = 0
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    for prec_name in ("fp32", "bf16"):
        nerf_amd.set_precision(prec_name)
        P = ops.current_precision()
        gate = 2e-5 if prec_name == "fp32" else 2e-2
        dt = torch.float32
        prop, mip = ProposalNetwork(10, 256), MipNeRF(10, 4, 256)
        prop.load_state_dict(W.proposal_state("he")); mip.load_state_dict(W.mip_state("he"))
        prop, mip = prop.cuda().train(), mip.cuda().train()
        gen = torch.Generator().manual_seed(5)
        pts3 = (torch.rand(M, 3, generator=gen) * 2 - 1).cuda()
        pts6 = torch.cat((pts3, torch.randn(M, 3, generator=gen).cuda()), -1).contiguous()
        cast = (lambda t: t.to(torch.bfloat16).float()) if prec_name == "bf16" else (lambda t: t)
        # ---------------- proposal
        wl = [l.weight.detach() for l in prop._linear_layers()]
        dens, dump = ops.proposal_forward_train(prop.packed(P), P, pts3)
        g = torch.randn(M, generator=gen).cuda()
        delta = ops.proposal_backward_chain(prop.packed_backward(P), P, g, dump)
        acts = [ops.train_dump_rows(dump, ops.NET_PROPOSAL, P, M, l, 256).float() for l in range(4)]
        enc = ops.train_dump_rows(dump, ops.NET_PROPOSAL, P, M, 4, 64).float()
        d = (cast(g)[:, None] * cast(wl[4])) * (acts[3] > 0)
        want_d = {3: d}
        for l in (3, 2, 1):
            d = (cast(d) @ cast(wl[l])) * (acts[l - 1] > 0)
            want_d[l - 1] = d
        for l in (3, 2, 1, 0):
            got = ops.train_dump_rows(delta, ops.NET_PROPOSAL, P, M, l, 256).float()
            r = rel(got, cast(want_d[l]))
            print("%s proposal delta_%d rel %.2e" % (prec_name, l, r)); bad += r > gate
        gW, gb = ops.proposal_weight_grads(P, M, dump, delta)
        dl = {l: ops.train_dump_rows(delta, ops.NET_PROPOSAL, P, M, l, 256).float() for l in range(4)}
        for l in (1, 2, 3):
            r = rel(gW[l], dl[l].t() @ acts[l - 1]); print("%s proposal dW%d rel %.2e" % (prec_name, l, r)); bad += r > gate
            r = rel(gb[l], dl[l].sum(0)); print("%s proposal db%d rel %.2e" % (prec_name, l, r)); bad += r > gate
        # the dumped encoding is in SLOT order: compare with the reference-order encoding through the gradient itself
        ref_enc = cast(torch.cat((pts3, ab._pe(pts3, 10)), -1))
        r = rel(gW[0], dl[0].t() @ ref_enc); print("%s proposal dW0 rel %.2e" % (prec_name, r)); bad += r > max(gate, 1e-4)
        r = rel(gb[0], dl[0].sum(0)); print("%s proposal db0 rel %.2e" % (prec_name, r)); bad += r > gate
        r = rel(gW[4], (cast(g)[None, :] @ acts[3])); print("%s proposal dW4 rel %.2e" % (prec_name, r)); bad += r > gate
        r = rel(gb[4], cast(g).sum().reshape(1)); print("%s proposal db4 rel %.2e" % (prec_name, r)); bad += r > gate
        # ---------------- MipNeRF
        layers = mip._linear_layers()
        wm = [l.weight.detach() for l in layers]; bm = [l.bias.detach() for l in layers]
        rgbo, dump = ops.mip_forward_train(mip.packed(P), P, pts6)
        g4 = torch.randn(M, 4, generator=gen).cuda()
        delta = ops.mip_backward_chain(mip.packed_backward(P), P, g4, rgbo, dump)
        acts = [ops.train_dump_rows(dump, ops.NET_MIP, P, M, l, 256).float() for l in range(7)]
        c = ops.train_dump_rows(dump, ops.NET_MIP, P, M, 7, 128).float()
        rgb = rgbo[:, :3]
        dpre = cast(g4[:, :3] * (1 - rgb) * rgb)
        dsig = cast(g4[:, 3:4])
        dc = (dpre @ cast(wm[10])) * (c > 0)
        wfold = wm[9][:, :256] @ wm[7]
        d6 = (cast(dc) @ cast(wfold) + dsig * cast(wm[8])) * (acts[6] > 0)
        want = {7: dc, 6: d6}
        d = d6
        for l, wmat in ((5, wm[6]), (4, wm[5]), (3, wm[4][:, 63:]), (2, wm[3]), (1, wm[2]), (0, wm[1])):
            d = (cast(d) @ cast(wmat)) * (acts[l] > 0)
            want[l] = d
        dl = {}
        for l in (7, 6, 5, 4, 3, 2, 1, 0):
            dl[l] = ops.train_dump_rows(delta, ops.NET_MIP, P, M, l, 128 if l == 7 else 256).float()
            r = rel(dl[l], cast(want[l])); print("%s mip delta_%d rel %.2e" % (prec_name, l, r)); bad += r > gate
        gW, gb = ops.mip_weight_grads(P, M, dump, delta, wm, bm)
        ex = cast(torch.cat((pts6[:, :3], ab._pe(pts6[:, :3], 10)), -1))
        dn = pts6[:, 3:] / pts6[:, 3:].norm(dim=-1, keepdim=True)
        ed = cast(torch.cat((dn, ab._pe(dn, 4)), -1))
        checks = []
        for L in (1, 2, 3, 5, 6):
            checks.append(("dW%d" % L, gW[L], dl[L].t() @ acts[L - 1])); checks.append(("db%d" % L, gb[L], dl[L].sum(0)))
        checks.append(("dW4", gW[4], torch.cat((dl[4].t() @ ex, dl[4].t() @ acts[3]), 1))); checks.append(("db4", gb[4], dl[4].sum(0)))
        checks.append(("dW0", gW[0], dl[0].t() @ ex)); checks.append(("db0", gb[0], dl[0].sum(0)))
        bott = acts[6] @ wm[7].t() + bm[7]
        dbott = dl[7] @ wm[9][:, :256]
        checks.append(("dW9", gW[9], torch.cat((dl[7].t() @ bott, dl[7].t() @ ed), 1))); checks.append(("db9", gb[9], dl[7].sum(0)))
        checks.append(("dW7", gW[7], dbott.t() @ acts[6])); checks.append(("db7", gb[7], dbott.sum(0)))
        checks.append(("dW8", gW[8], dsig.t() @ acts[6])); checks.append(("db8", gb[8], dsig.sum(0)))
        checks.append(("dW10", gW[10], dpre.t() @ c)); checks.append(("db10", gb[10], dpre.sum(0)))
        for name, got, wnt in checks:
            r = rel(got, wnt)
            lim = max(gate, 1e-4) if name in ("dW0", "dW4", "dW9", "dW7", "db7") else gate
            print("%s mip %s rel %.2e" % (prec_name, name, r)); bad += r > lim
    print("FAILED checks: %d" % bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
