"""TEST INFRASTRUCTURE ONLY -- generates tests/golden/*.pt from the REAL reference classes.

Run in the build container (needs /root/reference):  python oracle/make_golden.py
It (1) checks oracle/lbc_oracle.py against the reference modules on identical seeded
weights/inputs (forward eval+train, BN running-stat updates, all parameter gradients of a
phase-1 step), (2) checks the loss restatements against the AST-extracted reference
CoordConverter/LocationLoss, and (3) writes the reference's outputs as small fixtures so
the same checks can run on the GPU box where /root/reference does not exist.
Weights are NOT stored: they are regenerated from the seed (lbc_oracle.make_state_dict)
and verified through a checksum stored in the fixture.
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import lbc_oracle as O   # noqa: E402
from oracle import ref_shim          # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def seeded_inputs(kind, n, seed, h=None, w=None):
    g = torch.Generator().manual_seed(seed)
    if kind == "image":
        h, w = h or 160, w or 384
        x = torch.randint(0, 256, (n, h, w, 3), generator=g, dtype=torch.uint8).permute(0, 3, 1, 2).float() / 255.0
    else:
        h, w = h or 192, w or 192
        x = (torch.rand((n, 7, h, w), generator=g) < 0.1).float()
    speed = torch.rand(n, generator=g) * 10
    cmd = torch.randint(1, 5, (n,), generator=g).float()
    return x, speed, cmd


def grad_summary(t, gen_seed=1234):
    g = torch.Generator().manual_seed(gen_seed)
    flat = t.detach().reshape(-1)
    idx = torch.randint(0, flat.numel(), (min(16, flat.numel()),), generator=g)
    return {"sum": float(flat.double().sum()), "abssum": float(flat.double().abs().sum()),
            "max": float(flat.abs().max()), "idx": idx, "val": flat[idx].clone()}


def main():
    assert ref_shim.available(), "/root/reference not present"
    os.makedirs(GOLD, exist_ok=True)
    torch.set_num_threads(8)
    out = {}
    layouts = {}
    for kind, backbone, seed in (("image", "resnet34", 11), ("birdview", "resnet18", 12)):
        ref = ref_shim.build(kind, backbone, all_branch=True)
        rsd = ref.state_dict()
        layout = [(k, list(v.shape), str(v.dtype)) for k, v in rsd.items()]
        assert [(k, tuple(s)) for k, s, _ in layout] == [(k, tuple(s)) for k, s in O.state_dict_layout(kind, backbone)], "layout mismatch"
        layouts["%s_%s" % (kind, backbone)] = layout
        sd = O.make_state_dict(kind, backbone, seed)
        ref.load_state_dict(sd, strict=True)
        x, speed, cmd = seeded_inputs(kind, 2, seed + 100)
        onehot = ref_shim.load_one_hot()(cmd)
        assert torch.equal(onehot, O.one_hot(cmd))
        case = {"seed": seed, "input_seed": seed + 100, "checksum": O.checksum(sd)}
        # eval forward
        ref.eval()
        with torch.no_grad():
            rp, rpa = ref(x, speed, onehot)
            sd_e = {k: v.clone() for k, v in sd.items()}
            op, opa = O.policy_forward(sd_e, kind, backbone, x, speed, onehot, False)
        print(kind, "eval  |oracle-ref|", (op - rp).abs().max().item(), (opa - rpa).abs().max().item())
        assert (opa - rpa).abs().max() < 1e-5
        case["eval_pred"] = rp.clone(); case["eval_preds"] = rpa.clone()
        # train forward (batch statistics + running stat update)
        ref.train()
        with torch.no_grad():
            rp, rpa = ref(x, speed, onehot)
            sd_t = {k: v.clone() for k, v in sd.items()}
            op, opa = O.policy_forward(sd_t, kind, backbone, x, speed, onehot, True)
        print(kind, "train |oracle-ref|", (opa - rpa).abs().max().item())
        assert (opa - rpa).abs().max() < 1e-5
        rsd2 = ref.state_dict()
        for k in rsd2:
            if "running" in k or "num_batches" in k:
                assert torch.allclose(rsd2[k].float(), sd_t[k].float(), atol=1e-6), k
        case["train_pred"] = rp.clone(); case["train_preds"] = rpa.clone()
        case["running"] = {k: rsd2[k].clone() for k in ("conv.bn1.running_mean", "conv.bn1.running_var", "deconv.0.running_mean",
                                                          "deconv.0.running_var", "location_pred.2.0.running_var",
                                                          "conv.layer4.0.downsample.1.running_mean", "conv.bn1.num_batches_tracked")}
        out["%s_%s" % (kind, backbone)] = case

    # ---- phase-1 step gradients from the real reference (student r34 + teacher r18) ----
    defs = ref_shim.extract_training_defs("train_image_phase1.py", ["CoordConverter", "LocationLoss"])
    conv1 = defs["CoordConverter"](w=384, h=160, fov=90, world_y=1.4, fixed_offset=4.0, device="cpu")
    crit1 = defs["LocationLoss"]()
    student = ref_shim.build("image", "resnet34", all_branch=True)
    teacher = ref_shim.build("birdview", "resnet18", all_branch=True)
    ssd = O.make_state_dict("image", "resnet34", 21)
    tsd = O.make_state_dict("birdview", "resnet18", 22)
    student.load_state_dict(ssd); teacher.load_state_dict(tsd)
    student.train(); teacher.eval()
    n = 2
    rgb, speed, cmd = seeded_inputs("image", n, 300)
    bv, _, _ = seeded_inputs("birdview", n, 301)
    onehot = O.one_hot(cmd)
    with torch.no_grad():
        _, teac_all = teacher(bv, speed, onehot)
    # keep the 1/y singularity of the unprojection away: use well-conditioned synthetic teacher targets too
    pred, pred_all = student(rgb, speed, onehot)
    loss = crit1(conv1(pred_all), teac_all)
    loss.mean().backward()
    # oracle on the same
    sp = O.as_params(ssd)
    oloss, opred, opred_all, oteac = O.phase1_step_loss(sp, {k: v.clone() for k, v in tsd.items()}, "resnet34", "resnet18", rgb, bv, speed, onehot)
    oloss.mean().backward()
    print("phase1 loss ref", loss.detach(), "oracle", oloss.detach())
    assert torch.allclose(loss, oloss, rtol=1e-4, atol=1e-5)
    worst = 0.0
    grads = {}
    for k, p in student.named_parameters():
        if p.grad is None:
            assert k.startswith("conv.fc"), k
            continue
        og = sp[k].grad
        rel = (og - p.grad).abs().max().item() / (p.grad.abs().max().item() + 1e-12)
        worst = max(worst, rel)
        grads[k] = grad_summary(p.grad)
    print("phase1 grads: worst rel-to-max |oracle-ref| =", worst)
    assert worst < 2e-3
    out["phase1_step"] = {"student_seed": 21, "teacher_seed": 22, "rgb_seed": 300, "bv_seed": 301, "n": n,
                          "student_checksum": O.checksum(ssd), "teacher_checksum": O.checksum(tsd),
                          "loss": loss.detach().clone(), "pred_all": pred_all.detach().clone(), "teacher_all": teac_all.clone(),
                          "grads": grads}

    # ---- loss restatements vs the reference classes on random well-conditioned inputs ----
    g = torch.Generator().manual_seed(5)
    cam = torch.rand(6, 4, 5, 2, generator=g) * 1.6 - 0.8
    cam[..., 1] = cam[..., 1].abs() * 0.8 + 0.15           # below the horizon, away from the 1/y pole
    teac = torch.rand(6, 4, 5, 2, generator=g) * 2 - 1
    cam_r = cam.clone().requires_grad_(True)
    ref_map = conv1(cam_r)
    ref_l = crit1(ref_map, teac)
    ref_l.mean().backward()
    cam_o = cam.clone().requires_grad_(True)
    o_l = O.phase1_loss(O.phase1_unproject(cam_o), teac)
    o_l.mean().backward()
    assert torch.allclose(ref_l, o_l, rtol=1e-5, atol=1e-6) and torch.allclose(cam_r.grad, cam_o.grad, rtol=1e-4, atol=1e-6)
    out["phase1_loss"] = {"cam": cam, "teacher": teac, "map": ref_map.detach().clone(), "loss": ref_l.detach().clone(),
                          "dcam": cam_r.grad.clone()}
    # ---- phase-0: the reference's CoordConverter (cv2.projectPoints restated, see ref_shim) + LocationLoss --------------
    defs0 = ref_shim.extract_training_defs("train_image_phase0.py", ["CoordConverter", "LocationLoss"])
    conv0 = defs0["CoordConverter"](w=384, h=160, fov=90, world_y=1.4, fixed_offset=4.0, device="cpu")
    crit0 = defs0["LocationLoss"](w=384, h=160, device="cpu")
    tmap = torch.rand(6, 5, 2, generator=g) * 2 - 1          # teacher map-space waypoints, incl. ones that clip at the image border
    img_ref = conv0(tmap)
    assert torch.allclose(img_ref, O.phase0_project(tmap), rtol=1e-6, atol=1e-4), (img_ref - O.phase0_project(tmap)).abs().max()
    pred0 = (torch.rand(6, 5, 2, generator=g) * 2 - 1).requires_grad_(True)
    l0 = crit0(pred0, img_ref)
    l0.mean().backward()
    pred0o = pred0.detach().clone().requires_grad_(True)
    l0o = O.phase0_loss(pred0o, O.phase0_project(tmap))
    l0o.mean().backward()
    assert torch.allclose(l0, l0o, rtol=1e-6, atol=1e-7) and torch.allclose(pred0.grad, pred0o.grad, rtol=1e-6, atol=1e-8)
    out["phase0_loss"] = {"teacher_map": tmap, "image_xy": img_ref.clone(), "pred": pred0.detach().clone(), "loss": l0.detach().clone(),
                          "dpred": pred0.grad.clone()}
    # ---- bird-view behaviour cloning: LocationLoss(choice='l1') of train_birdview.py:33-54 (its ctor calls .cuda()) -------
    with ref_shim._cuda_noop():
        critb = ref_shim.extract_training_defs("train_birdview.py", ["LocationLoss"])["LocationLoss"](w=192, h=192, choice="l1")
    gt = torch.rand(6, 5, 2, generator=g) * 192
    predb = (torch.rand(6, 5, 2, generator=g) * 2 - 1).requires_grad_(True)
    lb = critb(predb, gt)
    lb.mean().backward()
    predbo = predb.detach().clone().requires_grad_(True)
    lbo = O.birdview_loss(predbo, gt)
    lbo.mean().backward()
    assert torch.allclose(lb, lbo, rtol=1e-6, atol=1e-7) and torch.allclose(predb.grad, predbo.grad, rtol=1e-6, atol=1e-8)
    out["birdview_loss"] = {"gt": gt, "pred": predb.detach().clone(), "loss": lb.detach().clone(), "dpred": predb.grad.clone()}
    # ---- dataset geometry: world_to_pixel of bird_view/utils/datasets/image_lmdb.py:22-30 (the module itself needs lmdb/cv2) ----
    import ast as _ast
    src = open(ref_shim.REFERENCE_ROOT + "/bird_view/utils/datasets/image_lmdb.py").read()
    ns_w = {"np": np, "PIXELS_PER_METER": 5}
    for node in _ast.parse(src).body:
        if isinstance(node, _ast.FunctionDef) and node.name == "world_to_pixel":
            exec(compile(_ast.Module([node], []), "image_lmdb.py", "exec"), ns_w)
    rs = np.random.RandomState(9)
    wargs, wout = [], []
    for _ in range(16):
        ox, oy = rs.uniform(-200, 200, 2)
        ang = rs.uniform(-np.pi, np.pi)
        x, y = ox + rs.uniform(-30, 30), oy + rs.uniform(-30, 30)
        a = (x, y, ox, oy, np.cos(ang), np.sin(ang))
        wargs.append(torch.tensor(a, dtype=torch.float64))
        wout.append(torch.from_numpy(np.asarray(ns_w["world_to_pixel"](*a), dtype=np.float64)))
    out["world_to_pixel"] = {"args": wargs, "out": wout}
    # ---- phase-2 resampling weight + batch_aug repeat vs the reference functions (training/phase2_utils.py) ----
    import ast
    src = open(ref_shim.REFERENCE_ROOT + "/training/phase2_utils.py").read()
    ns = {"torch": torch}
    for node in ast.parse(src).body:
        if isinstance(node, ast.FunctionDef) and node.name in ("get_weight", "repeat"):
            exec(compile(ast.Module([node], []), "phase2_utils.py", "exec"), ns)
    pred_cam = cam[:, 0].clone()                       # (6,5,2) well-conditioned camera-space predictions
    teac_sel = teac[:, 0].clone()
    ref_w = ns["get_weight"](conv1(pred_cam) / (0.5 * 192) - 1.0, teac_sel)
    assert torch.allclose(ref_w, O.phase2_weight(pred_cam, teac_sel), rtol=1e-5, atol=1e-7)
    t = torch.arange(24.0).view(4, 3, 2)
    assert torch.equal(ns["repeat"](t, 3), O.repeat(t, 3)) and torch.equal(ns["repeat"](t, 2, 1), O.repeat(t, 2, 1))
    out["phase2_weight"] = {"pred_cam": pred_cam, "teacher": teac_sel, "weight": ref_w.clone()}
    torch.save(out, os.path.join(GOLD, "reference_outputs.pt"))
    with open(os.path.join(GOLD, "state_dict_layout.json"), "w") as f:
        json.dump(layouts, f)
    print("wrote", GOLD)


if __name__ == "__main__":
    main()
