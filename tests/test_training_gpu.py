"""SURVEY 8 f4, the backward half: gradients of the camera head's training-side twin (reference __forward_PlaneCamRefHead,
camera_net/camera_head.py:737-923) from the hand-written HIP kernels (csrc/refine_bwd.hip) + the f32 GEMM kernel (nopesac_amd/training.py)
against torch.autograd on the oracle's restatement (oracle/nopesac_oracle.py::ransac_refine_train, itself pinned against the imported
reference in training mode: oracle/gen_golden.py stage H), and the optimiser step against torch.optim."""
import math

import pytest
import torch

from tests import golden_inputs as GI
from tests.util import rel_err

pytestmark = pytest.mark.gpu


def _pad(t, nq):
    out = torch.zeros(nq, t.shape[1])
    out[: t.shape[0]] = t
    return out


def _batch(nq, ms, seeds):
    from oracle import nopesac_oracle as O
    cases = [GI.refine_case(nq, m, s) for m, s in zip(ms, seeds)]
    B = len(cases)
    A = torch.zeros(B, nq, nq)
    for b, c in enumerate(cases):
        A[b, : c["A"].shape[0], : c["A"].shape[1]] = c["A"]
    geo = []
    for c in cases:
        gl, mm = O.geo_sequence(c["planes1"], c["planes2"], c["A"], nq)
        gg, _ = O.geo_sequence(c["planes1"], c["planes2"], c["A"], nq, c["init_rot"], c["init_trans"])
        ga, _ = O.geo_sequence(c["planes1"], c["planes2"], c["A"], nq, c["init_rot"], torch.zeros(3))
        geo.append((gl, gg, (((gg[:, 0:1] * ga[:, 0:1]) >= 0).float() - 0.5) * 2.0, mm))
    st = lambda k: torch.stack([c[k] for c in cases])
    dev_in = {"A": A, "p1": torch.stack([_pad(c["planes1"], nq) for c in cases]), "p2": torch.stack([_pad(c["planes2"], nq) for c in cases]),
              "n1": torch.tensor([c["planes1"].shape[0] for c in cases], dtype=torch.int32),
              "n2": torch.tensor([c["planes2"].shape[0] for c in cases], dtype=torch.int32)}
    return cases, st, geo, dev_in, GI.gt_pose_case(B, seeds[0])


def _oracle_grads(sd, nq, st, geo, gt, tag, weight, names, loss_weights=None):
    from oracle import nopesac_oracle as O
    sdg = {k: (v.clone().double().requires_grad_(True) if k in names else v.double()) for k, v in sd.items() if torch.is_tensor(v) and v.is_floating_point()}
    tf, rf = st("trans_feat").double().requires_grad_(True), st("rot_feat").double().requires_grad_(True)
    it, ir = st("init_trans").double().requires_grad_(True), st("init_rot").double().requires_grad_(True)
    losses, _ = O.ransac_refine_train(sdg, tf, rf, torch.stack([g[1] for g in geo]).double(), torch.stack([g[0] for g in geo]).double(),
                                      torch.stack([g[2] for g in geo]).double(), [g[3] for g in geo], it, ir, gt.double(),
                                      O.OracleConfig(num_queries=nq, out_cam_type="soft"), suffix=tag, weight=weight)
    total = sum(v * (loss_weights or {}).get(k, 1.0) for k, v in losses.items())
    total.backward()
    return losses, {k: sdg[k].grad for k in names}, {"init_trans_feat": tf.grad, "init_rot_feat": rf.grad, "init_trans": it.grad, "init_rot": ir.grad}


@pytest.mark.parametrize("nq,ms,seeds,tag,weight", [(50, (7, 2, 32, 50, 1), (67, 62, 92, 110, 61), "initCamRef", 1.0),
                                                     (50, (32,), (92,), "initRecCamRef", 0.5),
                                                     (64, (33, 64), (133, 164), "initCamRef_Aux", 2.0)])
def test_refine_head_gradients_match_autograd_on_the_oracle(device, nq, ms, seeds, tag, weight):
    """Every parameter gradient of the refinement head (11 MLP / Linear stacks: 40 tensors) and the gradients of its feature / pose inputs,
    for the sum of the seven losses, against float64 autograd on the oracle: the batches of the forward fixtures (m from 1 to nq - the clamp
    of the renormalised scores is active in the m = 50 / 64 pairs, inactive in the m = 1, 2 ones)."""
    from nopesac_amd.synth import synth_state_dict
    from nopesac_amd.training import RefineTrainer
    sd = synth_state_dict(nq)
    cases, st, geo, di, gt = _batch(nq, ms, seeds)
    tr = RefineTrainer.from_state_dict(sd, nq, device)
    names = list(tr.params)
    assert len(names) == 40
    dv = lambda t: t.to(device)
    feats = {k: dv(st(k)).requires_grad_(True) for k in ("trans_feat", "rot_feat", "init_trans", "init_rot")}
    losses = tr.losses(dv(di["A"]), dv(di["p1"]), dv(di["p2"]), dv(di["n1"]), dv(di["n2"]), feats["init_trans"], feats["init_rot"], feats["trans_feat"],
                       feats["rot_feat"], dv(gt), suffix=tag, weight=weight)
    grads = tr.backward(losses)
    o_loss, o_grads, o_in = _oracle_grads(sd, nq, st, geo, gt, tag, weight, names)
    for k in o_loss:
        assert rel_err(losses[k].detach(), o_loss[k].float()) < 2e-4, (k, float(losses[k]), float(o_loss[k]))
    worst = 0.0
    for k in names:
        ref = o_grads[k].float()
        assert torch.isfinite(grads[k]).all(), k
        scale = float(ref.abs().max())
        err = float((grads[k].cpu() - ref).abs().max()) / max(scale, 1e-12)
        worst = max(worst, err)
        assert err < 2e-3, (k, err, scale)
    for k_mine, k_ref in (("init_trans_feat", "init_trans_feat"), ("init_rot_feat", "init_rot_feat"), ("init_trans", "init_trans"), ("init_rot", "init_rot")):
        ref = o_in[k_ref].float()
        err = float((tr.input_grads[k_mine].cpu() - ref).abs().max()) / max(float(ref.abs().max()), 1e-12)
        assert err < 2e-3, (k_mine, err)
    # run-to-run: the backward pass has no atomics
    losses2 = tr.losses(dv(di["A"]), dv(di["p1"]), dv(di["p2"]), dv(di["n1"]), dv(di["n2"]), feats["init_trans"], feats["init_rot"], feats["trans_feat"],
                        feats["rot_feat"], dv(gt), suffix=tag, weight=weight)
    grads2 = tr.backward(losses2)
    assert all(torch.equal(grads[k], grads2[k]) for k in names)


def test_refine_head_gradients_with_per_loss_weights(device):
    """Each of the seven losses alone (loss_weights one-hot): the seven backward paths are checked one by one - a sign or scale error in one
    of them would hide behind the others in the summed test."""
    from nopesac_amd.synth import synth_state_dict
    from nopesac_amd.training import RefineTrainer
    from nopesac_amd import ops
    nq, ms, seeds, tag = 50, (7, 32, 3), (67, 92, 63), "initCamRef"
    sd = synth_state_dict(nq)
    cases, st, geo, di, gt = _batch(nq, ms, seeds)
    tr = RefineTrainer.from_state_dict(sd, nq, device)
    names = list(tr.params)
    dv = lambda t: t.to(device)
    for nm in ops.PLANE_CAM_REF_LOSS_NAMES:
        lw = {"%s_%s" % (n, tag): (1.0 if n == nm else 0.0) for n in ops.PLANE_CAM_REF_LOSS_NAMES}
        losses = tr.losses(dv(di["A"]), dv(di["p1"]), dv(di["p2"]), dv(di["n1"]), dv(di["n2"]), dv(st("init_trans")), dv(st("init_rot")),
                           dv(st("trans_feat")), dv(st("rot_feat")), dv(gt), suffix=tag, weight=1.0)
        grads = tr.backward(losses, lw)
        _, o_grads, _ = _oracle_grads(sd, nq, st, geo, gt, tag, 1.0, names, lw)
        gmax = max(float(o_grads[k].abs().max()) for k in names)
        assert gmax > 0, nm
        for k in names:
            ref = o_grads[k].float()
            err = float((grads[k].cpu() - ref).abs().max())
            assert err < 2e-3 * max(float(ref.abs().max()), 1e-3 * gmax), (nm, k, err)


@pytest.mark.parametrize("optimizer", ["ADAMW", "SGD"])
def test_refine_head_training_steps_match_torch_optim(device, optimizer):
    """Five optimiser steps on one batch: the HIP AdamW / SGD kernels against torch.optim on the oracle (float64 autograd gradients), and
    the loss goes down - forward, backward and update together are a training loop for this stage."""
    from oracle import nopesac_oracle as O
    from nopesac_amd.synth import synth_state_dict
    from nopesac_amd.training import RefineTrainer
    nq, ms, seeds, tag = 50, (7, 32, 12, 5), (67, 92, 72, 65), "initCamRef"
    sd = synth_state_dict(nq)
    cases, st, geo, di, gt = _batch(nq, ms, seeds)
    tr = RefineTrainer.from_state_dict(sd, nq, device)
    names = list(tr.params)
    dv = lambda t: t.to(device)
    ref_p = {k: sd[k].clone().double().requires_grad_(True) for k in names}
    opt = (torch.optim.AdamW([ref_p[k] for k in names], lr=2e-4, weight_decay=0.01) if optimizer == "ADAMW"
           else torch.optim.SGD([ref_p[k] for k in names], lr=2e-3, momentum=0.9, weight_decay=1e-4))
    mine, theirs = [], []
    for it in range(5):
        losses = tr.losses(dv(di["A"]), dv(di["p1"]), dv(di["p2"]), dv(di["n1"]), dv(di["n2"]), dv(st("init_trans")), dv(st("init_rot")),
                           dv(st("trans_feat")), dv(st("rot_feat")), dv(gt), suffix=tag)
        tr.backward(losses)
        mine.append(float(sum(v.detach() for v in losses.values())))
        if optimizer == "ADAMW":
            tr.step(lr=2e-4, optimizer="ADAMW", weight_decay=0.01)
        else:
            tr.step(lr=2e-3, optimizer="SGD", weight_decay=1e-4, momentum=0.9)
        sdg = {k: (ref_p[k] if k in ref_p else v.double()) for k, v in sd.items() if torch.is_tensor(v) and v.is_floating_point()}
        opt.zero_grad()
        ol, _ = O.ransac_refine_train(sdg, st("trans_feat").double(), st("rot_feat").double(), torch.stack([g[1] for g in geo]).double(),
                                      torch.stack([g[0] for g in geo]).double(), torch.stack([g[2] for g in geo]).double(), [g[3] for g in geo],
                                      st("init_trans").double(), st("init_rot").double(), gt.double(), O.OracleConfig(num_queries=nq, out_cam_type="soft"), suffix=tag)
        tot = sum(ol.values())
        tot.backward()
        theirs.append(float(tot))
        opt.step()
    assert mine[-1] < mine[0], mine
    for a, b in zip(mine, theirs):
        assert abs(a - b) < 2e-3 * abs(b), (mine, theirs)
    for k in names:
        assert rel_err(tr.params[k].detach(), ref_p[k].detach().float()) < 2e-3, k


def test_trained_parameters_reach_the_inference_head(device):
    """write_back(): the inference model's refinement stage runs on the updated parameters (packed copies rebuilt)."""
    from nopesac_amd.training import RefineTrainer
    from tests.util import make_model
    model = make_model(device)
    head = model.camera_head_list[0]
    tr = RefineTrainer.from_head(head)
    assert len(tr.params) == 40
    with torch.no_grad():
        for p in tr.params.values():
            p.mul_(1.01)
    tr.write_back(head)
    k = "geo_encoder.layers.0.weight"
    assert torch.equal(head.raw(k), tr.params["camera_head_list.0." + k].detach())
