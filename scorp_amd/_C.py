"""ctypes binding of libscorp_gs.so (C ABI in include/scorp_gs.h).

There is no CPU or PyTorch fallback: if the HIP library is missing or fails to load, importing the
rasterizer raises.  Build it with `python -m scorp_amd.build` (or __graft_entry__.build()).
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SCORP_GS_LIB", os.path.join(_HERE, "libscorp_gs.so"))   # override: A/B builds of the library

c_float_p = ctypes.c_void_p  # raw device pointers travel as integers


class ScorpGs3dInputs(ctypes.Structure):
    _fields_ = [
        ("num_gaussians", ctypes.c_int32), ("sh_degree", ctypes.c_int32), ("sh_coeffs", ctypes.c_int32),
        ("image_width", ctypes.c_int32), ("image_height", ctypes.c_int32),
        ("tanfovx", ctypes.c_float), ("tanfovy", ctypes.c_float), ("scale_modifier", ctypes.c_float),
        ("prefiltered", ctypes.c_int32), ("debug", ctypes.c_int32),
        ("bg", c_float_p), ("viewmatrix", c_float_p), ("projmatrix", c_float_p), ("campos", c_float_p),
        ("means3D", c_float_p), ("shs", c_float_p), ("colors_precomp", c_float_p), ("opacities", c_float_p),
        ("scales", c_float_p), ("rotations", c_float_p), ("cov3D_precomp", c_float_p),
        ("shs_rest", c_float_p), ("raw_params", ctypes.c_int32), ("num_views", ctypes.c_int32),
    ]


class ScorpGs3dGrads(ctypes.Structure):
    _fields_ = [(n, c_float_p) for n in ("means3D", "means2D", "shs", "colors_precomp", "opacities", "scales",
                                         "rotations", "cov3D_precomp", "shs_rest")]


class ScorpGs3dTrainView(ctypes.Structure):
    _fields_ = [("inputs", ctypes.c_void_p), ("out_radii", ctypes.c_void_p), ("state", ctypes.c_void_p),
                ("state_bytes", ctypes.c_size_t), ("pairs", ctypes.c_void_p), ("capacity", ctypes.c_uint64),
                ("out_color", ctypes.c_void_p), ("out_depth_raw", ctypes.c_void_p), ("out_alpha", ctypes.c_void_p),
                ("out_depth", ctypes.c_void_p), ("out_visible", ctypes.c_void_p), ("gt", ctypes.c_void_p),
                ("mask", ctypes.c_void_p), ("lambda_dssim", ctypes.c_float), ("backward_flags", ctypes.c_uint32),
                ("out_loss3", ctypes.c_void_p), ("loss_workspace", ctypes.c_void_p), ("loss_workspace_bytes", ctypes.c_size_t),
                ("grad_color", ctypes.c_void_p), ("grads", ctypes.c_void_p), ("backward_scratch", ctypes.c_void_p),
                ("backward_scratch_bytes", ctypes.c_size_t), ("out_header", ctypes.c_void_p), ("adam", ctypes.c_void_p)]


class ScorpFusedAdam(ctypes.Structure):
    _fields_ = [("exp_avg", ctypes.c_void_p * 6), ("exp_avg_sq", ctypes.c_void_p * 6), ("lr", ctypes.c_float * 6),
                ("_pad", ctypes.c_float * 2), ("beta1", ctypes.c_double), ("beta2", ctypes.c_double), ("eps", ctypes.c_double),
                ("step", ctypes.c_int32), ("_pad2", ctypes.c_int32), ("skipped_counter", ctypes.c_void_p),
                ("max_radii2D", ctypes.c_void_p), ("xyz_gradient_accum", ctypes.c_void_p), ("denom", ctypes.c_void_p)]


class ScorpGs2dTrainView(ctypes.Structure):
    _fields_ = [("inputs", ctypes.c_void_p), ("out_radii", ctypes.c_void_p), ("state", ctypes.c_void_p),
                ("state_bytes", ctypes.c_size_t), ("pairs", ctypes.c_void_p), ("capacity", ctypes.c_uint64),
                ("out_color", ctypes.c_void_p), ("out_allmap", ctypes.c_void_p), ("gt", ctypes.c_void_p),
                ("mask", ctypes.c_void_p), ("rays_d", ctypes.c_void_p), ("rays_o", ctypes.c_void_p),
                ("lambda_dssim", ctypes.c_float), ("depth_ratio", ctypes.c_float), ("lambda_normal", ctypes.c_float),
                ("lambda_dist", ctypes.c_float), ("out_loss3", ctypes.c_void_p), ("out_reg2", ctypes.c_void_p),
                ("loss_workspace", ctypes.c_void_p), ("loss_workspace_bytes", ctypes.c_size_t),
                ("reg_workspace", ctypes.c_void_p), ("reg_workspace_bytes", ctypes.c_size_t),
                ("grad_color", ctypes.c_void_p), ("grad_allmap", ctypes.c_void_p), ("grads", ctypes.c_void_p),
                ("backward_scratch", ctypes.c_void_p), ("backward_scratch_bytes", ctypes.c_size_t),
                ("backward_flags", ctypes.c_uint32), ("adam", ctypes.c_void_p)]


class ScorpRowTensor(ctypes.Structure):
    _fields_ = [("src", c_float_p), ("dst", c_float_p), ("row_floats", ctypes.c_uint32), ("zero_if_fresh", ctypes.c_uint32)]


class ScorpAdamTensor(ctypes.Structure):
    _fields_ = [("param", c_float_p), ("grad", c_float_p), ("exp_avg", c_float_p), ("exp_avg_sq", c_float_p),
                ("numel", ctypes.c_uint64), ("lr", ctypes.c_float), ("_pad", ctypes.c_float)]


EXPORTS = [
    "scorp_version", "scorp_source_sha", "scorp_last_error", "scorp_gs3d_state_bytes", "scorp_gs3d_pairs_bytes",
    "scorp_gs3d_backward_scratch_bytes", "scorp_gs3d_backward_scratch_bytes_ex", "scorp_gs3d_preprocess", "scorp_gs3d_num_pairs", "scorp_gs3d_render",
    "scorp_gs3d_render_image", "scorp_gs3d_render_score",
    "scorp_gs3d_check_overflow", "scorp_gs3d_backward", "scorp_gs3d_backward_ex", "scorp_gs3d_debug_geom", "scorp_gs3d_debug_tiles", "scorp_gs3d_debug_work",
    "scorp_loss_workspace_bytes", "scorp_loss_l1_ssim_forward", "scorp_loss_l1_ssim_backward",
    "scorp_knn_dist2", "scorp_gaussians_transform", "scorp_adam_step", "scorp_adam_step_guarded", "scorp_adam_step_guarded_ex", "scorp_densification_stats", "scorp_densification_stats_ex", "scorp_gather_rows", "scorp_gs3d_render_tail", "scorp_gs3d_render_tail_backward",
    "scorp_gs3d_pose_score_accumulate",
    "scorp_gs2d_state_bytes", "scorp_gs2d_backward_scratch_bytes", "scorp_gs2d_preprocess", "scorp_gs2d_render",
    "scorp_gs2d_render_image",
    "scorp_gs2d_backward", "scorp_gs2d_backward_ex", "scorp_gs2d_backward_scratch_bytes_ex", "scorp_gs2d_debug_geom", "scorp_gs2d_debug_tiles", "scorp_gs2d_maps_forward",
    "scorp_gs2d_maps_backward", "scorp_gs2d_regularizers_workspace_bytes", "scorp_gs2d_regularizers_forward",
    "scorp_gs2d_regularizers_backward", "scorp_gs3d_train_view", "scorp_gs2d_train_view",
    "scorp_prof_enable", "scorp_prof_select", "scorp_prof_num_kernels", "scorp_prof_kernel_name", "scorp_prof_collect",
]

BACKWARD_EXACT_FP32 = 1   # scorp_gs3d_backward_ex flag (include/scorp_gs.h)
BACKWARD_SCRATCH_ZEROED = 2   # scorp_gs3d_backward_ex flag: the caller cleared the accumulator rows already
BACKWARD_DETERMINISTIC = 4    # scorp_gs3d_backward_ex flag: no float atomics (plain partial rows + an ordered per-Gaussian sum)

_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run `python -m scorp_amd.build` "
            "(needs /opt/rocm/bin/hipcc). scorp_amd has no CPU fallback.")
    # PyTorch-ROCm bundles its own libamdhip64; it must be the HIP runtime already in the process when this library
    # is loaded, or the kernels register with a second runtime that has no device ("no ROCm-capable device").
    import torch  # noqa: F401
    L = ctypes.CDLL(LIB_PATH)
    vp, u64, i32, sz = ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int32, ctypes.c_size_t
    L.scorp_version.restype = ctypes.c_int
    L.scorp_last_error.restype = ctypes.c_char_p
    L.scorp_source_sha.restype = ctypes.c_char_p
    L.scorp_gs3d_state_bytes.restype = sz
    L.scorp_gs3d_state_bytes.argtypes = [i32, i32, i32]
    L.scorp_gs3d_pairs_bytes.restype = sz
    L.scorp_gs3d_pairs_bytes.argtypes = [u64]
    L.scorp_gs3d_backward_scratch_bytes.restype = sz
    L.scorp_gs3d_backward_scratch_bytes.argtypes = [i32]
    L.scorp_gs3d_backward_scratch_bytes_ex.restype = sz
    L.scorp_gs3d_backward_scratch_bytes_ex.argtypes = [i32, i32, i32, u64, ctypes.c_uint32]
    L.scorp_gs3d_preprocess.argtypes = [ctypes.POINTER(ScorpGs3dInputs), vp, vp, sz, vp]
    L.scorp_gs3d_num_pairs.argtypes = [vp, vp, ctypes.POINTER(u64)]
    L.scorp_gs3d_render.argtypes = [ctypes.POINTER(ScorpGs3dInputs), vp, vp, u64, vp, vp, vp, vp]
    L.scorp_gs3d_render_image.argtypes = [ctypes.POINTER(ScorpGs3dInputs), vp, vp, u64, vp, vp, vp, vp]
    L.scorp_gs3d_render_score.argtypes = [ctypes.POINTER(ScorpGs3dInputs), vp, vp, u64, vp, vp, i32, ctypes.c_float, vp, vp]
    L.scorp_gs3d_check_overflow.argtypes = [vp, vp, ctypes.POINTER(u64)]
    L.scorp_gs3d_backward.argtypes = [ctypes.POINTER(ScorpGs3dInputs), vp, vp, u64, vp, vp, vp,
                                      ctypes.POINTER(ScorpGs3dGrads), vp, sz, vp]
    L.scorp_gs3d_backward_ex.argtypes = [ctypes.POINTER(ScorpGs3dInputs), vp, vp, u64, vp, vp, vp,
                                         ctypes.POINTER(ScorpGs3dGrads), vp, sz, ctypes.c_uint32, vp]
    L.scorp_gs3d_debug_geom.argtypes = [vp, i32, i32, i32, vp, vp, vp, vp, vp, vp]
    L.scorp_gs3d_debug_tiles.argtypes = [vp, vp, u64, i32, i32, i32, vp, vp, vp]
    L.scorp_gs3d_debug_work.argtypes = [vp, i32, i32, i32, ctypes.POINTER(u64 * 3), vp]
    L.scorp_loss_workspace_bytes.restype = sz
    L.scorp_loss_workspace_bytes.argtypes = [i32, i32, i32]
    L.scorp_loss_l1_ssim_forward.argtypes = [vp, vp, vp, i32, i32, i32, ctypes.c_float, vp, vp, sz, i32, vp]
    L.scorp_loss_l1_ssim_backward.argtypes = [vp, vp, vp, i32, i32, i32, ctypes.c_float, vp, vp, vp, vp]
    L.scorp_gs3d_render_tail.argtypes = [vp, vp, ctypes.c_int64, vp, i32, vp, vp, vp]
    L.scorp_gs3d_render_tail_backward.argtypes = [vp, vp, vp, ctypes.c_int64, vp, vp, vp]
    L.scorp_gs3d_pose_score_accumulate.argtypes = [vp, vp, vp, vp, ctypes.c_int64, ctypes.c_float, vp, vp]
    L.scorp_gs2d_state_bytes.restype = sz
    L.scorp_gs2d_state_bytes.argtypes = [i32, i32, i32]
    L.scorp_gs2d_backward_scratch_bytes.restype = sz
    L.scorp_gs2d_backward_scratch_bytes.argtypes = [i32]
    L.scorp_gs2d_preprocess.argtypes = [ctypes.POINTER(ScorpGs3dInputs), vp, vp, sz, vp]
    L.scorp_gs2d_render.argtypes = [ctypes.POINTER(ScorpGs3dInputs), vp, vp, u64, vp, vp, vp]
    L.scorp_gs2d_render_image.argtypes = [ctypes.POINTER(ScorpGs3dInputs), vp, vp, u64, vp, vp, vp]
    L.scorp_gs2d_backward.argtypes = [ctypes.POINTER(ScorpGs3dInputs), vp, vp, u64, vp, vp, ctypes.POINTER(ScorpGs3dGrads), vp, sz, vp]
    L.scorp_gs2d_backward_ex.argtypes = [ctypes.POINTER(ScorpGs3dInputs), vp, vp, u64, vp, vp, ctypes.POINTER(ScorpGs3dGrads), vp, sz,
                                         ctypes.c_uint32, vp]
    L.scorp_gs2d_backward_scratch_bytes_ex.restype = sz
    L.scorp_gs2d_backward_scratch_bytes_ex.argtypes = [i32, i32, i32, u64, ctypes.c_uint32]
    L.scorp_gs2d_debug_geom.argtypes = [vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp]
    L.scorp_gs2d_debug_tiles.argtypes = [vp, vp, u64, i32, i32, i32, vp, vp, vp]
    L.scorp_gs2d_maps_forward.argtypes = [i32, i32, vp, vp, vp, vp, ctypes.c_float, vp, vp, vp, vp, vp, vp]
    L.scorp_gs2d_maps_backward.argtypes = [i32, i32, vp, vp, vp, vp, ctypes.c_float, vp, vp, vp, vp, vp, vp, vp, vp]
    L.scorp_gs2d_regularizers_workspace_bytes.restype = sz
    L.scorp_gs2d_regularizers_workspace_bytes.argtypes = [i32, i32]
    L.scorp_gs2d_regularizers_forward.argtypes = [i32, i32, vp, vp, vp, vp, ctypes.c_float, ctypes.c_float, ctypes.c_float, vp, vp, sz, vp]
    L.scorp_gs2d_regularizers_backward.argtypes = [i32, i32, vp, vp, vp, vp, ctypes.c_float, ctypes.c_float, ctypes.c_float, vp, vp, vp]
    L.scorp_gs3d_train_view.argtypes = [ctypes.POINTER(ScorpGs3dTrainView), vp]
    L.scorp_gs2d_train_view.argtypes = [ctypes.POINTER(ScorpGs2dTrainView), vp]
    L.scorp_knn_dist2.argtypes = [vp, i32, vp, vp]
    L.scorp_gaussians_transform.argtypes = [vp, vp, vp, vp, i32, i32, i32, vp, vp]
    L.scorp_adam_step.argtypes = [ctypes.POINTER(ScorpAdamTensor), i32, ctypes.c_double, ctypes.c_double, ctypes.c_double, i32, vp]
    L.scorp_adam_step_guarded.argtypes = [ctypes.POINTER(ScorpAdamTensor), i32, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                          i32, vp, vp]
    L.scorp_adam_step_guarded_ex.argtypes = [ctypes.POINTER(ScorpAdamTensor), i32, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                             i32, vp, vp, vp]
    L.scorp_gather_rows.argtypes = [ctypes.POINTER(ScorpRowTensor), i32, vp, u64, vp]
    L.scorp_densification_stats.argtypes = [i32, vp, vp, vp, i32, vp, vp, vp, vp, vp]
    L.scorp_densification_stats_ex.argtypes = [i32, vp, vp, vp, i32, i32, vp, vp, vp, vp, vp]
    L.scorp_prof_enable.argtypes = [ctypes.c_int]
    L.scorp_prof_select.argtypes = [u64]
    L.scorp_prof_kernel_name.restype = ctypes.c_char_p
    L.scorp_prof_kernel_name.argtypes = [ctypes.c_int]
    L.scorp_prof_collect.argtypes = [vp, vp]
    _lib = L
    return L


def current_stream_ptr():
    """The current HIP stream of the current device as an integer.  torch.cuda.current_stream() builds a Stream object through
    three layers of device-index helpers (9 us a call, six calls per training iteration); the raw getter is 0.3 us."""
    import torch
    try:
        return int(torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))
    except AttributeError:   # (a torch build without the raw getters)
        return int(torch.cuda.current_stream().cuda_stream)


def prof_enable(on=True, only=None):
    """Start / stop hipEvent bracketing of the library's kernels; `only` = iterable of kernel names to bracket."""
    L = lib()
    mask = (1 << 64) - 1
    if only is not None:
        names = [L.scorp_prof_kernel_name(k).decode() for k in range(L.scorp_prof_num_kernels())]
        mask = sum(1 << names.index(n) for n in only)
    check(L.scorp_prof_select(mask), "scorp_prof_select")
    check(L.scorp_prof_enable(1 if on else 0), "scorp_prof_enable")


def prof_collect():
    """{kernel name: (total ms, launches)} of the kernels timed since prof_enable(True); synchronises."""
    L = lib()
    n = L.scorp_prof_num_kernels()
    ms = (ctypes.c_double * n)()
    cnt = (ctypes.c_uint64 * n)()
    check(L.scorp_prof_collect(ms, cnt), "scorp_prof_collect")
    return {L.scorp_prof_kernel_name(k).decode(): (ms[k], int(cnt[k])) for k in range(n)}


def check(code, what):
    if code != 0:
        msg = lib().scorp_last_error()
        raise RuntimeError(f"{what} failed ({code}): {msg.decode(errors='replace') if msg else ''}")
