"""Helpers shared by the -m gpu parity tests: drive the HIP engine directly through the C ABI wrappers."""
import numpy as np
import torch

from oracle import graph as og


def relerr(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def l2err(a, b):
    """Relative Frobenius error: robust to the isolated arg-max flips fp16 storage causes in max-type ops."""
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    return float((a - b).norm() / (b.norm() + 1e-12))


def h16(t):
    """Round an fp32 tensor to fp16 precision (the storage format of the HIP path)."""
    return t.half().float()


def load_filled(module, layer, seed):
    g = og.Graph([layer], [], layer.args.get("nc", 0), 0, "", [], {})
    sd = og.fill_state(og.state_layout(g), seed)
    module.load_state_dict({k[len("model.0."):]: v for k, v in sd.items()}, strict=True)
    return sd


def act_grad_tensor(a):
    return a.st.gbuf[..., a.c0:a.c0 + a.C].permute(0, 3, 1, 2).float()


def run_fwd_bwd(module, xs, gy=None, training=True):
    """Forward (+ backward when gy is given) of a HipModule through the engine tape.
    xs: list of NCHW fp32 CPU tensors.  Returns (y, [gx], rt)."""
    from ultralytics.nn.tasks import initialize_weights
    initialize_weights(module)
    module.cuda()
    module.train(training)
    rt = module._runtime(torch.device("cuda", 0))
    eng = rt.eng
    eng.training = training
    eng.tape = [] if gy is not None else None
    rt.pack_all(True)
    acts = [rt.to_act(x.cuda()) for x in xs]
    y = module.forward_act(acts[0] if len(acts) == 1 else acts)
    gxs = None
    if gy is not None:
        g = y._gbuf()
        g[..., y.c0:y.c0 + y.C].copy_(gy.cuda().permute(0, 2, 3, 1))
        y.st.gwritten.append((y.c0, y.c0 + y.C))
        for f in reversed(eng.tape):
            f()
        gxs = [act_grad_tensor(a).cpu() if (a.needs_grad and a.st.gbuf is not None) else None for a in acts]
    eng.tape = None
    torch.cuda.synchronize()
    return rt.to_tensor(y).float().cpu(), gxs, rt
