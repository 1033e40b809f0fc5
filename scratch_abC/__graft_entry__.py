"""Driver entry points: build() compiles every HIP source for gfx950 (and the test-side oracle
artefacts); smoke() runs one tiny invocation of the hot path on cuda:0 and checks it against the
oracle."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def build():
    csrc = os.path.join(ROOT, "learningbycheating_amd", "csrc")
    subprocess.check_call(["make", "-s", "-C", csrc, "-j", "8", "hip"])
    # test infrastructure: the CPU-emulated build of the same kernel sources (tests/emu)
    subprocess.check_call(["make", "-s", "-C", csrc, "-j", "8", "emu"])
    import learningbycheating_amd  # noqa: F401
    from learningbycheating_amd import _lib
    lib = _lib.load()
    assert lib.lbc_backend().decode() == "hip-gfx950"
    # the oracle is pure Python (torch CPU); importing it is its "build".  The reference is Python too, so there is no
    # oracle/_ref binary; fixtures from the real reference are committed under tests/golden (oracle/make_golden.py).
    from oracle import lbc_oracle  # noqa: F401


def _frozen_decisions(eng):
    """the ReLU masks and max-pool taps of the executor's last training forward, read back from its workspace (the `frozen`
    argument of oracle.policy_forward: both sides then differentiate the SAME piece of the piecewise-linear network)"""
    acts = eng.activations()
    to = lambda t: t.permute(0, 3, 1, 2).cpu()
    fz = {"conv.maxpool": to(acts["conv.maxpool"]).float() > 0, "conv.maxpool.idx": to(acts["conv.maxpool.idx"])}
    for name, t in acts.items():
        if name.endswith(".conv1") and name != "conv.conv1":
            p = name[:-len(".conv1")]
            sc, sh = acts[p + ".bn1.scale"].reshape(1, -1, 1, 1).cpu(), acts[p + ".bn1.shift"].reshape(1, -1, 1, 1).cpu()
            fz[p + ".bn1"] = (to(t).double() * sc.double() + sh.double()) > 0        # (the GPU evaluates relu(y * scale + shift) as one fma)
            fz[p] = to(acts[p]).float() > 0
        elif name.startswith("deconv."):
            fz[name] = to(t).float() > 0
    return fz


def smoke():
    """one tiny forward + backward of the flagship model on cuda:0 through the module API, checked against the oracle: waypoints
    <= 1e-3 (the north-star bar; measured ~1e-5) and EVERY parameter gradient <= 1e-3 of its tensor's largest entry against the
    float64 oracle on the executor's own ReLU / max-pool decisions (measured ~1e-4: a 0.5 % gradient bug fails it)"""
    import torch
    from learningbycheating_amd import _lib
    from learningbycheating_amd.bird_view.models import ImagePolicyModelSS
    from oracle import lbc_oracle as O
    assert torch.cuda.is_available(), "smoke() needs a ROCm GPU"
    assert _lib.backend() == "hip-gfx950"
    dev = torch.device("cuda", 0)
    sd = O.make_state_dict("image", "resnet34", 4)
    net = ImagePolicyModelSS("resnet34", all_branch=True)
    net.load_state_dict(sd)
    net.to(dev).train()
    g = torch.Generator().manual_seed(0)
    x = torch.rand((2, 3, 160, 384), generator=g)
    speed = torch.rand(2, generator=g) * 10
    cmd = O.one_hot(torch.tensor([2.0, 4.0]))
    pred, preds = net(x.to(dev), speed.to(dev), cmd.to(dev))
    torch.cuda.synchronize()
    frozen = _frozen_decisions(next(iter(net._engines.values())))
    (preds.sum() + pred.sum()).backward()
    torch.cuda.synchronize()
    sp = O.as_params({k: (v.double() if v.dtype.is_floating_point else v.clone()) for k, v in sd.items()})
    opred, opreds = O.policy_forward(sp, "image", "resnet34", x.double(), speed.double(), cmd.double(), True, frozen=frozen)
    (opreds.sum() + opred.sum()).backward()
    err = (preds.detach().cpu().double() - opreds.detach()).abs().max().item()
    assert err < 1e-3, "forward parity vs oracle: %g" % err
    worst, worst_name = 0.0, ""
    for k, p in net.named_parameters():
        if p.grad is None or (k.startswith("location_pred") and k.endswith("bias")):     # (conv.fc.* has no gradient; the head's biases cancel in the softmax)
            continue
        ref = sp[k].grad
        e = (p.grad.detach().cpu().double() - ref).abs().max().item() / (ref.abs().max().item() + 1e-30)
        if e > worst:
            worst, worst_name = e, k
    assert worst < 1e-3, "gradient parity vs oracle: %g (%s)" % (worst, worst_name)
    print("smoke ok: |pred - oracle| = %.2e, worst per-tensor gradient error (rel. to the tensor's largest entry, frozen decisions) = %.2e (%s)"
          % (err, worst, worst_name))


if __name__ == "__main__":
    build()
    print("build ok")
