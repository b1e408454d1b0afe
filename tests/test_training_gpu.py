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
    """Every parameter gradient of the refinement head (13 MLP / Linear stacks: 80 tensors) and the gradients of its feature / pose inputs,
    for the sum of the seven losses, against float64 autograd on the oracle: the batches of the forward fixtures (m from 1 to nq - the clamp
    of the renormalised scores is active in the m = 50 / 64 pairs, inactive in the m = 1, 2 ones)."""
    from nopesac_amd.synth import synth_state_dict
    from nopesac_amd.training import RefineTrainer
    sd = synth_state_dict(nq)
    cases, st, geo, di, gt = _batch(nq, ms, seeds)
    tr = RefineTrainer.from_state_dict(sd, nq, device)
    names = list(tr.params)
    assert len(names) == 80 and {k.split('.')[2] for k in names} == set(__import__('nopesac_amd.training', fromlist=['MLPS']).MLPS + __import__('nopesac_amd.training', fromlist=['LINEARS']).LINEARS)
    dv = lambda t: t.to(device)
    feats = {k: dv(st(k)).requires_grad_(True) for k in ("trans_feat", "rot_feat", "init_trans", "init_rot")}
    losses = tr.losses(dv(di["A"]), dv(di["p1"]), dv(di["p2"]), dv(di["n1"]), dv(di["n2"]), feats["init_trans"], feats["init_rot"], feats["trans_feat"],
                       feats["rot_feat"], dv(gt), suffix=tag, weight=weight)
    grads = tr.backward(losses)
    o_loss, o_grads, o_in = _oracle_grads(sd, nq, st, geo, gt, tag, weight, names)
    for k in o_loss:
        assert rel_err(losses[k].detach(), o_loss[k].float()) < 2e-4, (k, float(losses[k]), float(o_loss[k]))
    gmax = max(float(o_grads[k].abs().max()) for k in names)
    report = []
    for k in names:
        ref = o_grads[k].float()
        assert torch.isfinite(grads[k]).all(), k
        scale = float(ref.abs().max())
        # (some gradients are zero by construction - e.g. the bias in front of a softmax: its gradient is the sum of a softmax backward -
        #  so the error is measured against the tensor's own scale, floored at 1e-4 of the largest gradient of the head)
        err = float((grads[k].cpu() - ref).abs().max()) / max(scale, 1e-4 * gmax)
        report.append((err, k, scale))
    report.sort(reverse=True)
    assert report[0][0] < 2e-3, report[:6]
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
           else torch.optim.SGD([ref_p[k] for k in names], lr=2e-5, momentum=0.9, weight_decay=1e-4))
    mine, theirs, first_step_err = [], [], None
    for it in range(5):
        losses = tr.losses(dv(di["A"]), dv(di["p1"]), dv(di["p2"]), dv(di["n1"]), dv(di["n2"]), dv(st("init_trans")), dv(st("init_rot")),
                           dv(st("trans_feat")), dv(st("rot_feat")), dv(gt), suffix=tag)
        tr.backward(losses)
        mine.append(float(sum(v.detach() for v in losses.values())))
        if optimizer == "ADAMW":
            tr.step(lr=2e-4, optimizer="ADAMW", weight_decay=0.01)
        else:
            tr.step(lr=2e-5, optimizer="SGD", weight_decay=1e-4, momentum=0.9)
        sdg = {k: (ref_p[k] if k in ref_p else v.double()) for k, v in sd.items() if torch.is_tensor(v) and v.is_floating_point()}
        opt.zero_grad()
        ol, _ = O.ransac_refine_train(sdg, st("trans_feat").double(), st("rot_feat").double(), torch.stack([g[1] for g in geo]).double(),
                                      torch.stack([g[0] for g in geo]).double(), torch.stack([g[2] for g in geo]).double(), [g[3] for g in geo],
                                      st("init_trans").double(), st("init_rot").double(), gt.double(), O.OracleConfig(num_queries=nq, out_cam_type="soft"), suffix=tag)
        tot = sum(ol.values())
        tot.backward()
        theirs.append(float(tot))
        opt.step()
        if it == 0:       # the update rule itself: after ONE step from identical parameters and (to 1e-3) identical gradients
            first_step_err = max(float((tr.params[k].detach().cpu() - ref_p[k].detach().float()).abs().max()) for k in names)
    assert mine[-1] < mine[0], mine
    # Adam divides by the running gradient magnitude: parameters whose gradient is ~0 move by +-lr on rounding noise, and the index losses
    # switch hypotheses - the two trajectories (f32 kernels / f64 autograd) agree closely for two steps and drift apart afterwards
    assert abs(mine[0] - theirs[0]) < 1e-4 * abs(theirs[0]) and abs(mine[1] - theirs[1]) < 2e-3 * abs(theirs[1]), (mine, theirs)
    assert abs(mine[2] - theirs[2]) < 2e-2 * abs(theirs[2]), (mine, theirs)
    if optimizer == "SGD":                                    # (Adam's first step is lr * sign(g): not comparable where g is rounding noise)
        assert first_step_err < 1e-5, first_step_err


@pytest.mark.parametrize("optimizer", ["ADAMW", "SGD"])
def test_optimizer_step_kernels_match_torch_optim(device, optimizer):
    """nopesac_adamw_step / nopesac_sgd_step against torch.optim on IDENTICAL gradients (the update rules of train_NopeSAC.py:150-157),
    four steps, including weight decay, bias correction and the first-step momentum rule."""
    from nopesac_amd.training import RefineTrainer
    g = torch.Generator().manual_seed(3)
    p0 = {"camera_head_list.0.rots.weight": torch.randn(4, 256, generator=g), "camera_head_list.0.rots.bias": torch.randn(4, generator=g)}
    tr = RefineTrainer({k: v.to(device) for k, v in p0.items()}, 50)
    ref = {k: v.clone().requires_grad_(True) for k, v in p0.items()}
    opt = (torch.optim.AdamW(list(ref.values()), lr=1e-3, weight_decay=0.05, betas=(0.9, 0.99), eps=1e-8) if optimizer == "ADAMW"
           else torch.optim.SGD(list(ref.values()), lr=1e-2, momentum=0.9, weight_decay=1e-3))
    for it in range(4):
        for k in p0:
            gk = torch.randn(p0[k].shape, generator=g) * (10.0 ** (-it))
            tr.params[k].grad = gk.to(device)
            ref[k].grad = gk.clone()
        if optimizer == "ADAMW":
            tr.step(lr=1e-3, optimizer="ADAMW", weight_decay=0.05, betas=(0.9, 0.99), eps=1e-8)
        else:
            tr.step(lr=1e-2, optimizer="SGD", weight_decay=1e-3, momentum=0.9)
        opt.step()
        for k in p0:
            assert float((tr.params[k].detach().cpu() - ref[k].detach()).abs().max()) < 2e-6, (it, k)


def test_trained_parameters_reach_the_inference_head(device):
    """write_back(): the inference model's refinement stage runs on the updated parameters (packed copies rebuilt)."""
    from nopesac_amd.training import RefineTrainer
    from tests.util import make_model
    model = make_model(device)
    head = model.camera_head_list[0]
    tr = RefineTrainer.from_head(head)
    assert len(tr.params) == 80
    with torch.no_grad():
        for p in tr.params.values():
            p.mul_(1.01)
    try:
        tr.write_back(head)
        k = "geo_encoder.layers.0.weight"
        assert torch.equal(head.raw(k), tr.params["camera_head_list.0." + k].detach())
    finally:                                                  # (tests/util.make_model caches the model: hand it back with its checkpoint)
        from nopesac_amd.synth import synth_state_dict
        model.load_state_dict(synth_state_dict(50))


def _oracle_camera_head_train_like_the_reference(sd, c, nq, head, names, conv_feats=None):
    """The oracle's training-mode camera head with the reference's DETACH points (camera_head.py:694, :723 the AIM re-embeds detached poses;
    :354-365 the geometry sequences come from detached initial poses) - oracle.camera_head_train keeps those paths differentiable, which is
    irrelevant for its (forward-only) use but not for a gradient oracle.  float64, autograd."""
    from oracle import nopesac_oracle as O
    p = "camera_head_list.0"
    cfg = O.OracleConfig(num_queries=nq)
    sdg = {k: (v.clone().double().requires_grad_(True) if k in names else v.double()) for k, v in sd.items() if torch.is_tensor(v) and v.is_floating_point()}
    dd = lambda t: t.double()
    f1 = {k: dd(v) for k, v in c["feats1"].items()}
    f2 = {k: dd(v) for k, v in c["feats2"].items()}
    gt = dd(c["gt_pose"])
    losses = {}
    if conv_feats is None:
        trans0, rot0, tf0, rf0, _ = O.pixel_pose_net(sdg, f1, f2, p)
    else:
        # the frozen conv stacks' outputs as the HIP kernels produced them (f32): the trainable layers on top, camera_head.py:655-667
        yt, yr = (dd(t.cpu()) for t in conv_feats)
        lin = lambda x, n: torch.nn.functional.linear(x, sdg[f"{p}.{n}.weight"], sdg[f"{p}.{n}.bias"])
        tf0, rf0 = torch.relu(lin(yt, "fc_trans")), torch.relu(lin(yr, "fc_rots"))
        trans0, rot0 = lin(tf0, "trans"), torch.nn.functional.normalize(lin(rf0, "rots"), p=2, dim=1)
    l_t, l_r = O.camera_pose_loss(torch.cat((trans0, rot0), -1), gt)
    losses["loss_tran_pixelReg"], losses["loss_rot_pixelReg"] = l_t * head.initial_cam_weight, l_r * head.initial_cam_weight

    def rec(trans_in, rot_in, suffix):
        trans_in, rot_in = trans_in.detach(), rot_in.detach()
        sig = ((rot_in[:, 0:1] >= 0.0).double() - 0.5) * 2.0
        rec_t, rec_r, rec_tf, rec_rf = O.aim_reembed(sdg, trans_in, rot_in, p)
        losses["loss_rot" + suffix] = (torch.nn.functional.normalize(rot_in * sig, dim=1) - rec_r).norm(dim=1).mean()
        losses["loss_trans" + suffix] = ((trans_in + 1e-10) - rec_t).norm(dim=1).mean()
        return rec_t, rec_r, rec_tf, rec_rf

    rec_t, rec_r, rec_tf, rec_rf = rec(trans0, rot0, "_initCamRec")
    B = gt.shape[0]
    for sfx, pl1, pl2, AA, w in (("", c["gt_planes1"], c["gt_planes2"], c["gt_A"], head.plane_cam_weight),
                                 ("_Aux", c["planes1"], c["planes2"], c["A"], head.plane_cam_weight_predplane)):
        for name, it, ir, itf, irf in (("initCamRef", trans0, rot0, tf0, rf0), ("initRecCamRef", rec_t, rec_r, rec_tf, rec_rf)):
            gl, gg, sg, ms = [], [], [], []
            for b in range(B):
                l, m = O.geo_sequence(dd(pl1[b]), dd(pl2[b]), dd(AA[b]), nq)
                g_, _ = O.geo_sequence(dd(pl1[b]), dd(pl2[b]), dd(AA[b]), nq, ir[b].detach(), it[b].detach())
                a_, _ = O.geo_sequence(dd(pl1[b]), dd(pl2[b]), dd(AA[b]), nq, ir[b].detach(), torch.zeros(3, dtype=torch.float64))
                gl.append(l); gg.append(g_); ms.append(m)
                sg.append((((g_[:, 0:1] * a_[:, 0:1]) >= 0).double() - 0.5) * 2.0)
            ls, _ = O.ransac_refine_train(sdg, itf, irf, torch.stack(gg), torch.stack(gl), torch.stack(sg), ms, it, ir, gt, cfg, suffix=name + sfx, weight=w, p=p)
            losses.update(ls)
    rec(dd(c["rand_trans"]), dd(c["rand_rot"]), "_randCamRecLBS_N1")
    total = sum(losses.values())
    if names:
        total.backward()
    return losses, {k: sdg[k].grad for k in names}


def test_camera_head_training_gradients(device):
    """The whole training-mode camera head (34 losses: pixel pose, AIM reconstruction x 2, four refinement passes) differentiated with
    respect to every Linear layer of the head (108 tensors: FC + regressors of the pixel pose net, AIM, refinement head) - the shared
    `rots` / `trans` regressors collect gradients from eleven call sites - against float64 autograd on the oracle with the reference's
    detach points.  The conv stacks of the pixel pose net are constants here (no backward kernels)."""
    from nopesac_amd.synth import synth_state_dict
    from nopesac_amd.training import CameraHeadTrainer
    from tests.util import make_model, nhwc
    nq, ms = 50, (7, 2, 19)
    c = GI.camera_train_case(nq, ms, 80)
    B = len(ms)
    sd = synth_state_dict(nq)
    model = make_model(device)
    head = model.camera_head_list[0]
    tr = CameraHeadTrainer.from_head(head)
    names = list(tr.params)
    assert len(names) == 108
    feats = {k: torch.cat([nhwc(c["feats1"][k]), nhwc(c["feats2"][k])]).to(device) for k in ("res3", "res4", "res5")}
    d = lambda k: c[k].to(device)
    losses = tr.camera_head_losses(head, feats, B, d("gt_planes1"), d("gt_planes2"), d("n1"), d("n2"), d("gt_A"), d("gt_pose"), d("planes1"), d("planes2"),
                                   d("n1"), d("n2"), d("A"), d("rand_rot"), d("rand_trans"))
    grads = tr.backward(losses)
    torch.set_default_dtype(torch.float64)                    # (the oracle creates a few constants in the default dtype)
    try:
        o_loss, o_grads = _oracle_camera_head_train_like_the_reference(sd, c, nq, head, names, tr.conv_feats)
        o_full, _ = _oracle_camera_head_train_like_the_reference(sd, c, nq, head, [])          # convs recomputed by the oracle (float64)
    finally:
        torch.set_default_dtype(torch.float32)
    assert set(losses) == set(o_loss) and len(losses) == 34
    for k in o_loss:
        assert rel_err(losses[k].detach(), o_loss[k].float().detach()) < 3e-4, (k, float(losses[k]), float(o_loss[k]))
        # (against the oracle's own float64 conv stacks: the f32 conv kernels' rounding reaches the losses at the 1e-3 level)
        assert rel_err(losses[k].detach(), o_full[k].float().detach()) < 1e-2, (k, float(losses[k]), float(o_full[k]))
    gmax = max(float(o_grads[k].abs().max()) for k in names)
    report = []
    for k in names:
        ref = o_grads[k].float()
        assert torch.isfinite(grads[k]).all(), k
        report.append((float((grads[k].cpu() - ref).abs().max()) / max(float(ref.abs().max()), 1e-4 * gmax), k))
    report.sort(reverse=True)
    assert report[0][0] < 3e-3, report[:6]
    try:
        tr.step(lr=1e-4)
        tr.write_back(head)
        with torch.no_grad():
            l2, _, _ = head.forward_train(feats, B, d("gt_planes1"), d("gt_planes2"), d("n1"), d("n2"), d("gt_A"), d("gt_pose"), d("planes1"), d("planes2"),
                                          d("n1"), d("n2"), d("A"), d("rand_rot"), d("rand_trans"))
        assert all(torch.isfinite(v) for v in l2.values()) and float(sum(l2.values())) != float(sum(v.detach() for v in losses.values()))
    finally:                                                  # (tests/util.make_model caches the model: hand it back with its checkpoint)
        model.load_state_dict(sd)


def test_full_model_gradient_clipping_and_solver_step(device):
    """RefineTrainer.clip_grad_norm = torch.nn.utils.clip_grad_norm_ (the reference's FullModelGradientClippingOptimizer), and
    step_from_cfg reads the solver keys of the reference's config (OPTIMIZER / BASE_LR / WEIGHT_DECAY / CLIP_GRADIENTS)."""
    from nopesac_amd.config import get_cfg
    from nopesac_amd.training import RefineTrainer
    g = torch.Generator().manual_seed(9)
    p0 = {"camera_head_list.0.rots.weight": torch.randn(4, 256, generator=g), "camera_head_list.0.trans.weight": torch.randn(3, 256, generator=g)}
    grads = {k: torch.randn(v.shape, generator=g) * 3 for k, v in p0.items()}
    for max_norm in (0.5, 1e6):
        tr = RefineTrainer({k: v.to(device) for k, v in p0.items()}, 50)
        ref = {k: v.clone().requires_grad_(True) for k, v in p0.items()}
        for k in p0:
            tr.params[k].grad = grads[k].to(device)
            ref[k].grad = grads[k].clone()
        coef = tr.clip_grad_norm(max_norm)
        total = torch.nn.utils.clip_grad_norm_(list(ref.values()), max_norm)
        assert abs(float(coef) - min(1.0, max_norm / (float(total) + 1e-6))) < 1e-6
        for k in p0:
            assert rel_err(tr.params[k].grad.cpu(), ref[k].grad) < 1e-5
    cfg = get_cfg()
    cfg.merge_from_list(["SOLVER.OPTIMIZER", "ADAMW", "SOLVER.BASE_LR", 0.001, "SOLVER.WEIGHT_DECAY", 0.05, "SOLVER.CLIP_GRADIENTS.ENABLED", True,
                         "SOLVER.CLIP_GRADIENTS.CLIP_TYPE", "full_model", "SOLVER.CLIP_GRADIENTS.CLIP_VALUE", 0.01])
    tr = RefineTrainer({k: v.to(device) for k, v in p0.items()}, 50)
    ref = {k: v.clone().requires_grad_(True) for k, v in p0.items()}
    opt = torch.optim.AdamW(list(ref.values()), lr=0.001, weight_decay=0.05)
    for k in p0:
        tr.params[k].grad = grads[k].to(device)
        ref[k].grad = grads[k].clone()
    tr.step_from_cfg(cfg)
    torch.nn.utils.clip_grad_norm_(list(ref.values()), 0.01)
    opt.step()
    for k in p0:
        assert float((tr.params[k].detach().cpu() - ref[k].detach()).abs().max()) < 2e-6
