"""Runs the 3x3 stride-1 convolutions of maps 40 / 80 pixels wide through every epilogue the step uses and saves the results:
tests/test_gpu_kernels.py compares the files of two processes started with different DY_CONV_FW (the tile scheme is chosen once per
process).  usage: fw_worker.py <out.pt>"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "experiment-yolo_amd"), os.path.dirname(os.path.abspath(__file__))]
import torch  # noqa: E402

from ultralytics.hip import DY_BN_COPIES, DY_EPI_ACCUM, DY_EPI_BIAS, DY_EPI_SILU, DY_EPI_STATS, DY_EPI_STATS_ACC  # noqa: E402
from ultralytics.hip.engine import ConvSpec, Engine, Storage  # noqa: E402

CASES = [(64, 64, 40, 40), (32, 32, 37, 40), (128, 64, 11, 40), (64, 128, 9, 40), (32, 16, 10, 40), (32, 48, 6, 40), (32, 32, 7, 80),
         (64, 64, 5, 80), (128, 32, 3, 80)]  # cin, cout, H, W


def main(out):
    eng = Engine("cuda:0")
    res = {}
    for cin, cout, H, W in CASES:
        g = torch.Generator().manual_seed(cin * 7 + cout * 3 + H + W)
        N = 3
        w = (torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5).cuda()
        b = torch.randn(cout, generator=g).cuda()
        sp = ConvSpec("c", w, b, None, 3, 1, 0)
        eng.prepare_conv(sp)
        eng.pack(sp)
        x = Storage(eng, N, H, W, cin)
        x.buf.copy_(torch.randn(N, H, W, cin, generator=g).half())
        r = Storage(eng, N, H, W, cout)
        r.buf.copy_(torch.randn(N, H, W, cout, generator=g).half())
        key = f"{cin}_{cout}_{H}_{W}"

        def fwd(epi, bias=0, part=0, y=None):
            y = torch.zeros(N, H, W, cout, dtype=torch.float16, device="cuda") if y is None else y
            eng.call("dy_conv_forward", x.buf.data_ptr(), cin, sp.wpack.data_ptr(), bias, y.data_ptr(), cout, part, N, H, W, cin, cout, 3, 1, 1,
                     0, 0, epi, None)
            return y
        res[key + "/plain"] = fwd(0)
        acc = torch.zeros(DY_BN_COPIES * 2 * cout, dtype=torch.float64, device="cuda")
        res[key + "/stats_y"] = fwd(DY_EPI_STATS | DY_EPI_STATS_ACC, 0, acc.data_ptr())
        res[key + "/stats_acc"] = acc.view(DY_BN_COPIES, 2, cout).sum(0)
        res[key + "/bias_silu"] = fwd(DY_EPI_BIAS | DY_EPI_SILU, b.data_ptr())
        res[key + "/accum"] = fwd(DY_EPI_ACCUM, 0, 0, r.buf.clone())
        if eng.L.dy_conv_res_supported(cin, cout, 3, 1):
            y = torch.zeros(N, H, W, cout, dtype=torch.float16, device="cuda")
            eng.call("dy_conv_forward_res", x.buf.data_ptr(), cin, sp.wpack.data_ptr(), b.data_ptr(), r.buf.data_ptr(), cout, y.data_ptr(), cout,
                     N, H, W, cin, cout, 3, 1)
            res[key + "/res"] = y
        y32 = torch.zeros(N, H, W, cout, device="cuda")
        eng.conv_bias(sp, x.act(), y32.data_ptr(), cout, True)
        res[key + "/f32"] = y32
    torch.cuda.synchronize()
    torch.save({k: v.cpu() for k, v in res.items()}, out)


if __name__ == "__main__":
    main(sys.argv[1])
