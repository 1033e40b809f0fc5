"""The oracle (oracle/lbc_oracle.py) against the fixtures produced from the REAL reference classes
(oracle/make_golden.py -> tests/golden/): runs wherever the suite runs (no /root/reference needed)."""
import json
import os

import torch

from oracle import lbc_oracle as O
from oracle.make_golden import seeded_inputs

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_gold():
    return torch.load(os.path.join(GOLD, "reference_outputs.pt"))


def test_state_dict_layout_matches_reference():
    lay = json.load(open(os.path.join(GOLD, "state_dict_layout.json")))
    for name, (kind, backbone) in {"image_resnet34": ("image", "resnet34"), "birdview_resnet18": ("birdview", "resnet18")}.items():
        ref = [(k, tuple(s)) for k, s, _ in lay[name]]
        assert ref == [(k, tuple(s)) for k, s in O.state_dict_layout(kind, backbone)]
    assert len(lay["image_resnet34"]) == 275 and len(lay["birdview_resnet18"]) == 179


def test_forward_matches_reference_fixtures():
    gold = load_gold()
    for name, (kind, backbone) in {"image_resnet34": ("image", "resnet34"), "birdview_resnet18": ("birdview", "resnet18")}.items():
        c = gold[name]
        sd = O.make_state_dict(kind, backbone, c["seed"])
        assert abs(O.checksum(sd) - c["checksum"]) <= 1e-9 * abs(c["checksum"]), "seeded weights differ from the generating site"
        x, speed, cmd = seeded_inputs(kind, 2, c["input_seed"])
        onehot = O.one_hot(cmd)
        with torch.no_grad():
            p, pa = O.policy_forward({k: v.clone() for k, v in sd.items()}, kind, backbone, x, speed, onehot, False)
            assert (pa - c["eval_preds"]).abs().max() < 1e-5 and (p - c["eval_pred"]).abs().max() < 1e-5
            sdt = {k: v.clone() for k, v in sd.items()}
            p, pa = O.policy_forward(sdt, kind, backbone, x, speed, onehot, True)
            assert (pa - c["train_preds"]).abs().max() < 1e-5
            for k, v in c["running"].items():
                assert torch.allclose(sdt[k].float(), v.float(), atol=1e-6), k


def test_phase1_loss_and_gradient_match_reference_fixture():
    g = load_gold()["phase1_loss"]
    cam = g["cam"].clone().requires_grad_(True)
    m = O.phase1_unproject(cam)
    loss = O.phase1_loss(m, g["teacher"])
    loss.mean().backward()
    assert torch.allclose(m, g["map"], rtol=1e-5, atol=1e-4)
    assert torch.allclose(loss, g["loss"], rtol=1e-5, atol=1e-6)
    assert torch.allclose(cam.grad, g["dcam"], rtol=1e-4, atol=1e-6)
    # closed form of SURVEY 8(a)#13-14: m_x = 0.175 x/y, m_y = 1.20833 - 0.175/y (normalised map coordinates)
    x, y = g["cam"][..., 0], g["cam"][..., 1]
    assert torch.allclose(m[..., 0] / 96 - 1, 0.175 * x / y, atol=2e-5)
    assert torch.allclose(m[..., 1] / 96 - 1, 1.2083333 - 0.175 / y, atol=2e-5)


def test_phase1_step_gradients_match_reference_fixture():
    g = load_gold()["phase1_step"]
    ssd = O.make_state_dict("image", "resnet34", g["student_seed"])
    tsd = O.make_state_dict("birdview", "resnet18", g["teacher_seed"])
    assert abs(O.checksum(ssd) - g["student_checksum"]) <= 1e-9 * abs(g["student_checksum"])
    rgb, speed, cmd = seeded_inputs("image", g["n"], g["rgb_seed"])
    bv, _, _ = seeded_inputs("birdview", g["n"], g["bv_seed"])
    sp = O.as_params(ssd)
    loss, _, pred_all, teac = O.phase1_step_loss(sp, tsd, "resnet34", "resnet18", rgb, bv, speed, O.one_hot(cmd))
    loss.mean().backward()
    assert torch.allclose(loss, g["loss"], rtol=1e-4)
    assert (pred_all - g["pred_all"]).abs().max() < 1e-5
    for k, s in g["grads"].items():
        got = sp[k].grad.reshape(-1)[s["idx"]]
        assert torch.allclose(got, s["val"], rtol=1e-3, atol=1e-4 * s["max"] + 1e-9), k
    assert sp["conv.fc.weight"].grad is None


def test_known_answers():
    # SpatialSoftmax corner check (the commented self-test at reference common.py:192-201, restated for 40x96 and 48x48)
    for (h, w) in ((40, 96), (48, 48)):
        px, py = O.softmax_positions(h, w)
        for (i, j) in ((h - 1, 0), (h - 1, w // 2), (h - 1, w - 1), (0, w // 2)):
            f = torch.zeros(1, 1, h, w)
            f[0, 0, i, j] = 100.0
            out = O.spatial_softmax(f, px, py)[0, 0]
            assert abs(out[0].item() - (-1 + 2 * j / (w - 1))) < 1e-6 and abs(out[1].item() - (-1 + 2 * i / (h - 1))) < 1e-6
    # select_branch with a one-hot command picks exactly that branch
    b = torch.randn(3, 4, 5, 2)
    oh = O.one_hot(torch.tensor([1.0, 4.0, 2.0]))
    s = O.select_branch(b, oh)
    assert torch.equal(s[0], b[0, 0]) and torch.equal(s[1], b[1, 3]) and torch.equal(s[2], b[2, 1])
    # one_hot clamps out-of-range commands (train_utils.py:36)
    assert torch.equal(O.one_hot(torch.tensor([0.0, 7.0])), torch.tensor([[1.0, 0, 0, 0], [0, 0, 0, 1.0]]))
    # phase-0 projection: a point straight ahead at 10 m lands on the image centre column, below the horizon
    t = torch.zeros(1, 5, 2)
    t[..., 1] = 1 - 2 * ((10 - 4.0) * 5) / 192.0
    uv = O.phase0_project(t)
    assert torch.allclose(uv[..., 0], torch.full((1, 5), 192.0)) and torch.allclose(uv[..., 1], torch.full((1, 5), 80 + 192 * 1.4 / 10))


def test_phase0_and_birdview_losses_match_reference_fixtures():
    """fixtures = outputs of the reference's own CoordConverter / LocationLoss classes (AST-extracted from
    training/train_image_phase0.py:36-89 and train_birdview.py:33-54 by oracle/make_golden.py)"""
    g = load_gold()
    p0 = g["phase0_loss"]
    assert torch.allclose(O.phase0_project(p0["teacher_map"]), p0["image_xy"], rtol=1e-6, atol=1e-4)
    pred = p0["pred"].clone().requires_grad_(True)
    loss = O.phase0_loss(pred, O.phase0_project(p0["teacher_map"]))
    loss.mean().backward()
    assert torch.allclose(loss, p0["loss"], rtol=1e-6, atol=1e-7) and torch.allclose(pred.grad, p0["dpred"], rtol=1e-6, atol=1e-8)
    bv = g["birdview_loss"]
    pred = bv["pred"].clone().requires_grad_(True)
    loss = O.birdview_loss(pred, bv["gt"])
    loss.mean().backward()
    assert torch.allclose(loss, bv["loss"], rtol=1e-6, atol=1e-7) and torch.allclose(pred.grad, bv["dpred"], rtol=1e-6, atol=1e-8)


def test_phase2_weight_and_repeat_match_reference_fixture():
    g = load_gold()["phase2_weight"]
    assert torch.allclose(O.phase2_weight(g["pred_cam"], g["teacher"]), g["weight"], rtol=1e-5, atol=1e-7)
    t = torch.arange(6.0).view(3, 2)
    assert torch.equal(O.repeat(t, 2), torch.tensor([[0., 1.], [0., 1.], [2., 3.], [2., 3.], [4., 5.], [4., 5.]]))
