#!/usr/bin/env python3
"""Diagnostic of the hand-written Ref-NeRF training kernels against torch.autograd of the reference expression (tests/torch_spec.py ref_expr)
on the device: training forward == inference forward, density gradients (RefNeRF.get_grad) of both networks, every parameter gradient."""
import sys

import torch
import torch.nn.functional as F

ROOT = __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
import nerf_amd
import weights as W
from nerf_amd import ops
sys.path.insert(0, __import__('os').path.join(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))), 'tests'))
import torch_spec as ab                                 # the ops' torch specifications (test infrastructure)
from nerf_amd.addtional import ProposalNetwork
from nerf_amd.ref_model import RefNeRF


def rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
    tag = sys.argv[2] if len(sys.argv) > 2 else "he"
    bad = 0
    gen = torch.Generator().manual_seed(3)
    pos = (torch.rand(M, 3, generator=gen) * 2 - 1).cuda()
    dirs = F.normalize(torch.randn(M, 3, generator=gen), dim=-1).cuda()
    noise = (torch.randn(M, 128, generator=gen) * 0.1).cuda()
    g_out = torch.randn(M, 7, generator=gen).cuda()
    for prec_name in ("fp32", "bf16"):
        nerf_amd.set_precision(prec_name)
        P = ops.current_precision()
        gate = 2e-4 if prec_name == "fp32" else 6e-2
        # density gradients: a single ReLU unit an ulp from zero flips between the kernel's and torch's fp32 forward (1e-3 on the whole
        # tensor); in bf16 the 2^f factors of the encoding's derivative amplify the operand rounding -- judged by the direction
        dgate = 5e-3 if prec_name == "fp32" else 0.25
        cosdir = lambda a, b: F.cosine_similarity(a, b, dim=-1).mean().item()
        net = RefNeRF(10, 4); net.load_state_dict(W.ref_state(tag)); net = net.cuda().train()
        prop = ProposalNetwork(10, 256); prop.load_state_dict(W.proposal_state(tag)); prop = prop.cuda().train()
        pts6 = torch.cat((pos, dirs), -1).contiguous()
        with torch.no_grad():
            rgbo, normal, dump, aux = ops.ref_forward_train(net.packed(P), P, pts6, noise)
            rgbo2, normal2 = ops.ref_forward(net.packed(P), P, pts6, noise=noise)
        print("%s train fwd == fwd: %s %s" % (prec_name, torch.equal(rgbo, rgbo2), torch.equal(normal, normal2))); bad += not torch.equal(rgbo, rgbo2)
        names = [n for n, _ in net.named_parameters()]
        leaves = {n: p.detach().clone().requires_grad_(True) for n, p in net.named_parameters()}
        x = pos.detach().clone().requires_grad_(True)
        y = ab.ref_expr(x, dirs, noise, leaves, net.integrated_dir_enc)                      # (M,7) fp32 reference expression
        print("%s fwd vs expr: rgbo %.2e normal %.2e" % (prec_name, rel(torch.cat((rgbo, normal), -1), y.detach()), rel(normal, y.detach()[:, 4:])))
        if prec_name == "fp32":                               # every dumped activation against the expression's intermediates
            with torch.no_grad():
                Pm = {n: p.detach() for n, p in net.named_parameters()}
                lin = lambda nm, t: F.linear(t, Pm[nm + ".weight"], Pm[nm + ".bias"])
                ex = torch.cat((pos, ab._pe(pos, 10)), -1)
                acts = {}
                hcur = ex
                for i, l in enumerate((0, 2, 4, 6)):
                    hcur = F.relu(lin("spa_block1.%d" % l, hcur)); acts[i] = hcur
                gcur = torch.cat((ex, hcur), -1)
                for i, l in enumerate((0, 2, 4, 6)):
                    gcur = F.relu(lin("spa_block2.%d" % l, gcur)); acts[4 + i] = gcur
                nrm, dif, tint = lin("norm_col_tint_head", gcur).split((3, 3, 3), -1)
                rough, dens_ = lin("rho_tau_head", gcur).split((1, 1), -1)
                rough = F.softplus(rough - 1.0)
                bvec = lin("bottle_neck", gcur) + noise
                nn_ = -nrm / (nrm.norm(dim=-1, keepdim=True) + 1e-7)
                refl = dirs - 2.0 * torch.sum(dirs * nn_, -1, keepdim=True) * nn_
                allin = torch.cat((bvec, net.integrated_dir_enc(refl, rough), torch.sum(nn_ * dirs, -1, keepdim=True)), -1)
                rcur = allin
                for i, l in enumerate((0, 2, 4, 6)):
                    rcur = F.relu(lin("dir_block1.%d" % l, rcur)); acts[9 + i] = rcur
                rcur = torch.cat((allin, rcur), -1)
                for i, l in enumerate((0, 2, 4, 6)):
                    rcur = F.relu(lin("dir_block2.%d" % l, rcur)); acts[13 + i] = rcur
                for slot in sorted(acts):
                    got = ops.train_dump_rows(dump, ops.NET_REF, P, M, slot, 256).float()
                    print("   dump slot %2d rel %.2e  mask flips %d" % (slot, rel(got, acts[slot]), int(((got > 0) != (acts[slot] > 0)).sum())))
                bn_got = ops.train_dump_rows(dump, ops.NET_REF, P, M, 8, 128).float()
                print("   dump bottle-neck rel %.2e" % rel(bn_got, bvec))
        # density gradient (get_grad) of Ref-NeRF
        gx_ref, = torch.autograd.grad(y[:, 3].sum(), x, retain_graph=True)
        blob = net.packed_backward(P)
        gx = ops.density_grad(ops.NET_REF, blob, P, dump, pts6)
        r = rel(gx, gx_ref); print("%s ref density grad rel %.2e  mean direction cosine %.5f" % (prec_name, r, cosdir(gx, gx_ref))); bad += r > dgate or cosdir(gx, gx_ref) < 0.98
        sc = torch.rand(M, generator=gen).cuda()
        r = rel(ops.density_grad(ops.NET_REF, blob, P, dump, pts6, scale=sc), gx_ref * sc[:, None]); print("%s   scaled rel %.2e" % (prec_name, r)); bad += r > dgate
        # proposal density gradient
        xp = pos.detach().clone().requires_grad_(True)
        lw = [l.weight.detach() for l in prop._linear_layers()]; lb = [l.bias.detach() for l in prop._linear_layers()]
        yp = ab.proposal_expr(xp, lw, lb)
        gp_ref, = torch.autograd.grad(yp.sum(), xp)
        _, dump_p = ops.proposal_forward_train(prop.packed(P), P, pos)
        gp = ops.density_grad(ops.NET_PROPOSAL, prop.packed_backward(P), P, dump_p, pos)
        r = rel(gp, gp_ref); print("%s proposal density grad rel %.2e  mean direction cosine %.5f" % (prec_name, r, cosdir(gp, gp_ref))); bad += r > dgate or cosdir(gp, gp_ref) < 0.98
        # parameter gradients.  Reference: autograd of the same expression with the ReLU decisions of the KERNEL's forward (masks from the
        # dump): one unit whose pre-activation is an ulp from zero flips between two fp32 evaluations and moves a whole gradient by
        # 1e-3, which says nothing about the backward under test.
        def masked_expr(Pm):
            lin = lambda nm, t: F.linear(t, Pm[nm + ".weight"], Pm[nm + ".bias"])
            mask = lambda slot: (ops.train_dump_rows(dump, ops.NET_REF, P, M, slot, 256).float() > 0).float()
            ex = torch.cat((pos, ab._pe(pos, 10)), -1)
            hcur = ex
            for i, l in enumerate((0, 2, 4, 6)):
                hcur = lin("spa_block1.%d" % l, hcur) * mask(i)
            gcur = torch.cat((ex, hcur), -1)
            for i, l in enumerate((0, 2, 4, 6)):
                gcur = lin("spa_block2.%d" % l, gcur) * mask(4 + i)
            nrm, dif, tint = lin("norm_col_tint_head", gcur).split((3, 3, 3), -1)
            rough, dens_ = lin("rho_tau_head", gcur).split((1, 1), -1)
            rough = F.softplus(rough - 1.0)
            bvec = lin("bottle_neck", gcur) + noise
            nn_ = -nrm / (nrm.norm(dim=-1, keepdim=True) + 1e-7)
            refl = dirs - 2.0 * torch.sum(dirs * nn_, -1, keepdim=True) * nn_
            allin = torch.cat((bvec, net.integrated_dir_enc(refl, rough), torch.sum(nn_ * dirs, -1, keepdim=True)), -1)
            rcur = allin
            for i, l in enumerate((0, 2, 4, 6)):
                rcur = lin("dir_block1.%d" % l, rcur) * mask(9 + i)
            rcur = torch.cat((allin, rcur), -1)
            for i, l in enumerate((0, 2, 4, 6)):
                rcur = lin("dir_block2.%d" % l, rcur) * mask(13 + i)
            rgb = torch.sigmoid(lin("spec_rgb_head.0", rcur)) * torch.sigmoid(tint) + torch.sigmoid(dif)
            return torch.cat((rgb, dens_, nn_), -1)
        if prec_name == "fp32":
            want = torch.autograd.grad(masked_expr(leaves), [leaves[n] for n in names], g_out, allow_unused=True)
        else:
            want = torch.autograd.grad(y, [leaves[n] for n in names], g_out, allow_unused=True)
        gw, gb = ops.ref_backward(blob, P, dump, aux, dirs, g_out, net._ide_table(pos.device))
        got = RefNeRF._grads_by_name(gw, gb)
        worst = 0.0
        for n, wnt in zip(names, want):
            r = rel(got[n], wnt)
            # bf16: the reference expression runs in fp32 on other ReLU masks; only the direction is meaningful (see tests)
            cos = F.cosine_similarity(got[n].reshape(1, -1).double(), wnt.reshape(1, -1).double()).item()
            flag = (r > gate) if prec_name == "fp32" else (cos < 0.95)
            if flag or n.endswith("weight"):
                print("%s %-28s rel %.2e cos %.5f%s" % (prec_name, n, r, cos, "  <-- BAD" if flag else ""))
            bad += flag
            worst = max(worst, r)
        print("%s worst parameter-gradient rel %.2e" % (prec_name, worst))
    print("FAILED checks: %d" % bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
