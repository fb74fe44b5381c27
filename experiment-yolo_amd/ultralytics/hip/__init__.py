"""ctypes binding of libdealyolo_hip.so (the C ABI declared in include/dealyolo_hip.h).

The product path has NO fallback: if the library is missing or a kernel returns an error code, a RuntimeError is
raised.  PyTorch is used only for device memory, streams and torch.distributed.
"""
from __future__ import annotations

import ctypes as C
import os
import sys

# hipGraph safety on this ROCm (7.2): with the runtime's "graph packet capture" (AQL packets and kernel arguments pre-built when a
# graph is instantiated) a burst of ordinary launches between two replays -- ~1,000 of this library's kernels: six eval-mode
# forwards, or the traces of two other launch lists -- overwrites the captured arguments of an ALREADY INSTANTIATED graph: its
# next replays compute with garbage (non-finite gradients) while the same launch list issued eagerly is fine (DESIGN.md
# section 14, profiles/r02_graph_packet_capture.txt).  DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 turns the pre-building off (no measured
# cost: 14.21 vs 14.23 ms per step) and the corruption does not occur.  The flag is read when HIP initialises, so it is set here,
# at import; a process that initialised the GPU earlier without it gets no graphs (GRAPH_SAFE, checked by StepPlan).
_FLAG = "DEBUG_CLR_GRAPH_PACKET_CAPTURE"


def _hip_already_up():
    t = sys.modules.get("torch")
    try:
        return bool(t is not None and t.cuda.is_initialized())
    except Exception:
        return False


if os.environ.get(_FLAG) is None and not _hip_already_up():
    os.environ[_FLAG] = "0"  # inherited by child processes (torch.distributed.run ranks)
GRAPH_SAFE = os.environ.get(_FLAG) == "0"  # unset + HIP already initialised, or set to something else by the user: no graphs

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DY_HIP_LIB") or os.path.normpath(os.path.join(_HERE, "..", "..", "csrc", "libdealyolo_hip.so"))

DY_EPI_STATS, DY_EPI_BIAS, DY_EPI_SILU, DY_EPI_F32OUT, DY_EPI_ACCUM, DY_EPI_STATS_ACC = 1, 2, 4, 8, 16, 32
DY_BN_COPIES = 16  # include/dealyolo_hip.h
DY_ACT_NONE, DY_ACT_SILU, DY_ACT_LEAKY = 0, 1, 2
_ERR = {-1: "DY_ERR_ARG (unsupported shape/argument)", -2: "DY_ERR_LAUNCH (HIP launch failed)",
        -3: "DY_ERR_ALIGN (pointer/stride alignment)"}

vp, i32, f32, i64, sz = C.c_void_p, C.c_int, C.c_float, C.c_long, C.c_size_t
ip = C.POINTER(C.c_int)
lp = C.POINTER(C.c_long)


class DyLossArgs(C.Structure):
    _fields_ = [("nl", i32), ("B", i32), ("nc", i32), ("ncp", i32), ("nmax", i32),
                ("box", vp * 4), ("cls", vp * 4), ("dbox", vp * 4), ("dcls", vp * 4),
                ("H", i32 * 4), ("W", i32 * 4), ("stride", f32 * 4),
                ("t_batch_idx", vp), ("t_cls", vp), ("t_boxes", vp), ("n_targets", i32), ("n_targets_dev", vp),
                ("img_w", f32), ("img_h", f32), ("hyp_box", f32), ("hyp_cls", f32), ("hyp_dfl", f32),
                ("use_wiou", i32), ("use_nwd", i32), ("iou_ratio", f32), ("gscale", vp), ("scalars", vp),
                ("workspace", vp), ("dbox_rows_only", i32), ("box_from_input", i32), ("box_in", vp * 4), ("box_in_ld", i32 * 4),
                ("box_w", vp * 4), ("box_b", vp * 4), ("box_in_coef", vp * 4)]


class DySegs(C.Structure):
    """include/dealyolo_hip.h DySegs: a channel concatenation that is never materialised (hip/engine.py, SegAct)."""
    _fields_ = [("nseg", i32), ("c_end", i32 * 8), ("ld", i32 * 8), ("acc", i32 * 8), ("ptr", vp * 8)]


# name -> (restype, argtypes); every exported symbol of include/dealyolo_hip.h appears here (tests/test_abi.py)
SIGNATURES = {
    "dy_abi_version": (i32, []),
    "dy_loss_args_bytes": (i32, []),
    "dy_conv_geometry": (i32, [i32, i32, i32, i32, ip, ip, ip, ip, ip, ip, ip, ip]),
    "dy_pack_weights": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "dy_pack_desc_bytes": (i32, []),
    "dy_pack_desc_fill": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32]),
    "dy_pack_weights_batched": (i32, [vp, i32, i32, vp]),
    "dy_conv_forward": (i32, [vp, i32, vp, vp, vp, i32, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, ip, vp]),
    "dy_segs_bytes": (i32, []),
    "dy_conv1x1_segs_supported": (i32, [i32, i32, C.POINTER(DySegs)]),
    "dy_conv1x1_forward_segs": (i32, [C.POINTER(DySegs), vp, vp, vp, i32, vp, i32, i32, i32, i32, i32, i32, vp]),
    "dy_conv1x1_input_grad_segs": (i32, [vp, i32, vp, C.POINTER(DySegs), i32, i32, i32, i32, i32, vp]),
    "dy_conv1x1_wgrad_bn_segs": (i32, [C.POINTER(DySegs), vp, i32, vp, i32, vp, vp, vp, vp, vp, f32, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "dy_conv_res_supported": (i32, [i32, i32, i32, i32]),
    "dy_conv_forward_res": (i32, [vp, i32, vp, vp, vp, i32, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
    "dy_conv_red_supported": (i32, [i32, i32, i32]),
    "dy_conv_input_grad_red": (i32, [vp, i32, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp, i32, vp, vp, i32, vp]),
    "dy_conv_kernel_name": (i32, [i32, i32, i32, i32, C.c_char_p, i32]),
    "dy_conv1x1_segs_kernel_name": (i32, [i32, i32, C.POINTER(DySegs), C.c_char_p, i32]),
    "dy_conv_kernel_name_at": (i32, [i32, i32, i32, i32, i32, i32, i32, C.c_char_p, i32]),
    "dy_wgrad_kernel_name": (i32, [i32, i32, i32, i32, C.c_char_p, i32]),
    "dy_wgrad_kernel_name_at": (i32, [i32, i32, i32, i32, i32, i32, i32, C.c_char_p, i32]),
    "dy_wgrad_reduce_desc_bytes": (i32, []),
    "dy_wgrad_reduce_desc_fill": (i32, [vp, vp, i32, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32]),
    "dy_wgrad_reduce_batched": (i32, [vp, i32, i32, vp]),
    "dy_match_predictions": (i32, [vp, vp, vp, vp, vp, i32, vp, vp, i32, i32, i32, i32, vp, vp, vp, vp]),
    "dy_box_iou": (i32, [vp, i32, vp, i32, vp, vp]),
    "dy_conv_num_partials": (i32, [i32, i32, i32, i32, i32, i32, i32, i32]),
    "dy_wgrad_workspace": (i32, [i32, i32, i32, i32, i32, i32, i32, ip, lp]),
    "dy_conv_wgrad": (i32, [vp, i32, vp, i32, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
    "dy_stem_grid": (i32, [i32, i32, i32]),
    "dy_stem_forward": (i32, [vp, vp, vp, i32, vp, i32, i32, i32, f32, vp]),
    "dy_stem_forward_eval": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, f32, i32, vp]),
    "dy_stem_wgrad_bn": (i32, [vp, vp, i32, vp, i32, vp, vp, vp, vp, f32, vp, i32, i32, i32, f32, vp]),
    "dy_conv_wgrad_bias": (i32, [vp, i32, vp, i32, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
    "dy_wgrad_reduce_desc_bias": (i32, [vp, vp, vp, i32]),
    "dy_conv_wgrad_bn": (i32, [vp, i32, vp, i32, vp, i32, vp, vp, vp, vp, vp, f32, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
    "dy_conv_wgrad_ld_bn": (i32, [vp, i32, vp, i32, vp, i32, vp, vp, vp, vp, vp, f32, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
    "dy_ldconv_sample": (i32, [vp, i32, vp, i32, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
    "dy_ldconv_sample_backward": (i32, [vp, i32, vp, i32, vp, vp, i32, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
    "dy_ldconv_sample_backward_gather": (i32, [vp, i32, vp, i32, vp, vp, i32, vp, i32, i32, vp, vp, i32, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
    "dy_f32_to_f16_add": (i32, [vp, vp, i32, i64, i32, i32, vp]),
    "dy_pack_weights_ld": (i32, [vp, vp, i32, i32, i32, i32, i32, vp]),
    "dy_conv_wgrad_ld": (i32, [vp, i32, vp, i32, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
    "dy_bn_finalize": (i32, [vp, i32, f32, vp, i32, f32, vp, i32, f32, vp, vp, vp, vp, vp, i32, f32, f32, f32, i32, vp]),
    "dy_bn_eval_coef": (i32, [vp, vp, vp, vp, vp, i32, f32, vp]),
    "dy_bn_act_apply": (i32, [vp, i32, vp, i32, vp, i32, vp, i64, i32, i32, vp]),
    "dy_bn_act_apply_acc": (i32, [vp, i32, vp, i32, vp, i32, vp, vp, vp, vp, vp, vp, i64, i32, i32, f32, f32, f32, vp]),
    "dy_bn_act_bwd_reduce_acc": (i32, [vp, i32, vp, i32, vp, vp, i64, i32, i32, vp, i32, i32, vp]),
    "dy_bn_act_bwd_apply_acc": (i32, [vp, i32, vp, i32, vp, i32, vp, vp, vp, vp, i64, i32, i32, f32, vp]),
    "dy_bn_act_bwd_reduce": (i32, [vp, i32, vp, i32, vp, vp, i32, i64, i32, i32, ip, vp]),
    "dy_bn_bwd_finalize": (i32, [vp, i32, vp, vp, vp, i32, f32, i32, vp]),
    "dy_bn_act_bwd_apply": (i32, [vp, i32, vp, i32, vp, i32, vp, vp, i64, i32, i32, i32, vp]),
    "dy_import_image": (i32, [vp, vp, i32, i32, i32, i32, i32, f32, vp]),
    "dy_crop_letterbox_u8": (i32, [vp, i32, i32, vp, vp, i32, i32, vp, vp]),
    "dy_refine_select": (i32, [vp, vp, vp, vp, vp, i32, C.c_float, C.c_float, vp, vp, vp]),
    "dy_nms_hard": (i32, [vp, vp, vp, i32, C.c_float, vp, vp]),
    "dy_warp_import_u8": (i32, [vp, vp, vp, i32, i32, i32, vp]),
    "dy_warp_slot_bytes": (i32, []),
    "dy_import_image_u8": (i32, [vp, vp, i32, i32, i32, i32, vp, vp, vp, vp]),
    "dy_add": (i32, [vp, i32, vp, i32, vp, i32, vp, i32, i64, i32, vp]),
    "dy_upsample2x": (i32, [vp, i32, vp, i32, i32, i32, i32, i32, i32, i32, vp]),
    "dy_maxpool5": (i32, [vp, i32, vp, i32, vp, i32, i32, i32, i32, vp]),
    "dy_maxpool5_backward": (i32, [vp, i32, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "dy_bn_group_max": (i32, []),
    "dy_bn_act_apply_acc_split": (i32, [vp, i32, vp, i32, vp, i32, i32, vp, vp, vp, vp, vp, vp, i64, i32, i32, f32, f32, f32, vp]),
    "dy_bn_act_bwd_reduce_acc_split": (i32, [vp, i32, vp, i32, i32, vp, i32, vp, vp, i64, i32, i32, vp]),
    "dy_conv1x1_wgrad_bn_planes": (i32, [C.POINTER(DySegs), vp, i32, vp, vp, i32, i32, vp, i32, vp, vp, vp, vp, vp, f32, vp, vp, i32, i32, i32,
                                         i32, i32, i32, vp]),
    "dy_bn_act_apply_acc_group": (i32, [i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "dy_bn_act_bwd_reduce_acc_group": (i32, [i32, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "dy_bn_act_bwd_reduce_rows": (i32, [vp, i32, vp, i32, vp, vp, i32, i32, i32, i32, vp, i32, i32, vp]),
    "dy_head_box_decode": (i32, [vp, i32, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]),
    "dy_cls_head_supported": (i32, [i32, i32]),
    "dy_cls_head_slabs": (i32, []),
    "dy_cls_head_forward": (i32, [vp, i32, vp, vp, vp, vp, i64, i32, i32, vp]),
    "dy_cls_head_backward": (i32, [vp, i32, vp, vp, vp, vp, i32, i32, vp, vp, i64, i32, i32, vp]),
    "dy_head_box_decode_levels": (i32, [i32, vp, vp, vp, vp, vp, vp, i32, vp, i32, vp, vp, i32, i32, vp]),
    "dy_conv1x1_rows_backward_levels": (i32, [i32, vp, vp, vp, vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, i32, vp, vp, i32, i32, vp]),
    "dy_cls_head_forward_levels": (i32, [i32, vp, vp, vp, vp, vp, vp, vp, i32, i32, vp]),
    "dy_cls_head_backward_levels": (i32, [i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, vp]),
    "dy_conv1x1_rows_supported": (i32, [i32, i32]),
    "dy_conv1x1_rows_slabs": (i32, [i32, i32, i32]),
    "dy_conv1x1_rows_backward": (i32, [vp, i32, vp, vp, i32, vp, i32, i32, vp, vp, i32, i32, vp, vp, i32, i32, i32, i32, i32, vp]),
    "dy_sppf_pool3_supported": (i32, [i32, i32, i32]),
    "dy_sppf_pool3": (i32, [vp, i32, i32, vp, vp, vp, i32, i32, i32, vp]),
    "dy_sppf_pool3_backward": (i32, [vp, i32, i32, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "dy_scalseq_tail": (i32, [vp, i32, vp, i32, vp, i32, vp, i32, vp, i32, vp, i32, i32, i32, i32, vp]),
    "dy_scalseq_tail_backward": (i32, [vp, i32, vp, i32, vp, i32, vp, i32, vp, i32, vp, vp, vp, i32, i32, i32, i32, i32,
                                       i32, i32, ip, vp]),
    "dy_scalseq_tail_backward_all": (i32, [vp, i32, vp, i32, vp, i32, vp, i32, vp, i32, vp, i32, vp, i32, vp, vp, vp, i32, i32,
                                           i32, i32, i32, i32, ip, vp]),
    "dy_zoom_pool": (i32, [vp, i32, vp, i32, i32, i32, i32, i32, vp]),
    "dy_zoom_pool_backward": (i32, [vp, i32, vp, i32, vp, i32, i32, i32, i32, i32, i32, vp]),
    "dy_copy_slice": (i32, [vp, i32, vp, i32, i64, i32, vp]),
    "dy_fill_zero": (i32, [vp, sz, vp]),
    "dy_loss_workspace_bytes": (sz, [i32, i32, i32]),
    "dy_loss_workspace_layout": (i32, [i32, i32, i32, C.POINTER(sz), C.POINTER(sz), C.POINTER(sz)]),
    "dy_detection_loss": (i32, [C.POINTER(DyLossArgs), vp]),
    "dy_tal_assign": (i32, [vp, ip, ip, C.POINTER(f32), i32, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp]),
    "dy_head_infer_supported": (i32, [i32, i32, i32, i32]),
    "dy_head_infer_levels": (i32, [i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp, vp]),
    "dy_decode_predictions": (i32, [vp, vp, ip, ip, C.POINTER(f32), i32, i32, i32, i32, vp, vp]),
    "dy_nms_candidates": (i32, [vp, i32, i32, i32, f32, i32, vp, i32, vp, vp, vp, vp, i32, vp]),
    "dy_nms_presort_workspace": (sz, [i32, i32]),
    "dy_nms_presort": (i32, [vp, vp, vp, vp, i32, i32, i32, vp, vp, vp, vp, vp]),
    "dy_soft_nms": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, f32, f32, f32, f32, vp]),
    "dy_set_hyper": (i32, [vp, vp, vp]),
    "dy_optimizer_step": (i32, [vp, vp, vp, vp, vp, i64, i64, i64, vp, vp, vp, i64, vp, vp, vp, i32, vp]),
    "dy_optimizer_step_seg": (i32, [vp, vp, vp, vp, vp, i64, lp, vp, vp, vp, i64, vp, vp, vp, i32, vp]),
    "dy_axpy_f32": (i32, [vp, vp, f32, i64, vp]),
}

_LIB = None


def lib():
    """Load (once) and return the shared library; raises if it has not been built."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found. The DEAL-YOLO hot path has no CPU/PyTorch fallback: build the HIP library "
                f"first (python -c 'import __graft_entry__ as g; g.build()' or `make -C experiment-yolo_amd/csrc`).")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        if L.dy_segs_bytes() != C.sizeof(DySegs):
            raise RuntimeError(f"{LIB_PATH} lays DySegs out in {L.dy_segs_bytes()} bytes, this binding in {C.sizeof(DySegs)}: rebuild the library")
        if L.dy_loss_args_bytes() != C.sizeof(DyLossArgs):  # a stale .so (or a stale binding) would read garbage pointers
            raise RuntimeError(f"{LIB_PATH} was built with a DyLossArgs of {L.dy_loss_args_bytes()} bytes, this binding lays out "
                               f"{C.sizeof(DyLossArgs)}: rebuild the library (make -C experiment-yolo_amd/csrc)")
        _LIB = L
    return _LIB


def check(rc, name):
    if rc != 0:
        raise RuntimeError(f"{name} failed: {_ERR.get(rc, rc)}")
