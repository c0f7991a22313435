"""ctypes binding of libvita_hip.so (include/vita_hip.h).

The product path has no CPU fallback: if the shared library is missing or does not export a
declared symbol, importing an operator raises.  Build it with `python __graft_entry__.py`.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# VITA_AMD_LIB: load another build of the same library (profiling experiments, e.g. profiles/ablate_ps.sh)
LIB_PATH = os.environ.get("VITA_AMD_LIB") or os.path.join(_HERE, "lib", "libvita_hip.so")

VH_ACT_NONE, VH_ACT_GELU, VH_ACT_RELU, VH_ACT_SILU = 0, 1, 2, 3

c_void_p, c_int, c_long, c_float, c_size_t = C.c_void_p, C.c_int, C.c_long, C.c_float, C.c_size_t


class GemmArgs(C.Structure):
    _fields_ = [
        ("A", c_void_p), ("lda", c_long), ("a_rows", c_int), ("a_rowidx", c_void_p),
        ("nseg", c_int), ("seglen", c_int), ("segrow", c_int * 16),
        ("W", c_void_p), ("W_up", c_void_p), ("ldw", c_long), ("w_group_stride", c_long),
        ("group_off", c_void_p), ("ngroups", c_int),
        ("C", c_void_p), ("ldc", c_long), ("c_rowidx", c_void_p),
        ("bias", c_void_p), ("scale", c_void_p), ("resid", c_void_p), ("ldr", c_long),
        ("M", c_int), ("N", c_int), ("K", c_int), ("act", c_int),
        ("ws", c_void_p), ("ws_bytes", C.c_size_t), ("ksplit", c_int),
    ]


class GemmPsArgs(C.Structure):
    _fields_ = [
        ("A_hi", c_void_p), ("A_lo", c_void_p), ("lda", c_long), ("a_rowidx", c_void_p),
        ("W", c_void_p), ("W_up", c_void_p), ("ldw", c_long), ("w_group_stride", c_long),
        ("group_off", c_void_p), ("ngroups", c_int),
        ("C", c_void_p), ("ldc", c_long), ("C_hi", c_void_p), ("C_lo", c_void_p), ("ldc_split", c_long),
        ("c_rowidx", c_void_p), ("bias", c_void_p), ("scale", c_void_p), ("resid", c_void_p), ("ldr", c_long),
        ("M", c_int), ("N", c_int), ("K", c_int), ("act", c_int), ("wide", c_int),
        ("ksplit", c_int), ("c_split_stride", c_long), ("nslab_out", c_void_p),
    ]


class AttnArgs(C.Structure):
    _fields_ = [
        ("Q", c_void_p), ("ldq", c_long), ("hsq", c_long),
        ("K", c_void_p), ("ldk", c_long), ("hsk", c_long),
        ("V", c_void_p), ("ldv", c_long), ("hsv", c_long),
        ("P", c_void_p), ("ldp", c_long), ("hsp", c_long),
        ("bias_u", c_void_p), ("bias_v", c_void_p),
        ("O", c_void_p), ("ldo", c_long),
        ("bsq", c_long), ("bsk", c_long), ("bso", c_long),
        ("B", c_int), ("Hq", c_int), ("Hkv", c_int), ("Sq", c_int), ("Sk", c_int), ("d", c_int),
        ("causal", c_int), ("q_off", c_int), ("klen", c_int), ("chunk", c_int), ("left", c_int),
        ("scale", c_float),
    ]


class EncoderLayerArgs(C.Structure):
    _fields_ = [
        ("x", c_void_p), ("h_in", c_void_p), ("h_out", c_void_p),
        ("M", c_int), ("C", c_int), ("F", c_int), ("heads", c_int), ("B", c_int),
        ("qkv_w", c_void_p), ("qkv_b", c_void_p),
        ("proj_w", c_void_p), ("proj_b", c_void_p), ("ls1", c_void_p),
        ("n2_w", c_void_p), ("n2_b", c_void_p),
        ("fc1_w", c_void_p), ("fc1_b", c_void_p),
        ("fc2_w", c_void_p), ("fc2_b", c_void_p), ("ls2", c_void_p),
        ("next_w", c_void_p), ("next_b", c_void_p),
        ("act", c_int), ("eps", c_float),
        ("P", c_void_p), ("ldp", c_long), ("bias_u", c_void_p), ("bias_v", c_void_p), ("klen", c_int), ("chunk", c_int), ("left", c_int),
        ("qkv", c_void_p), ("attn", c_void_p), ("hmid", c_void_p), ("mid", c_void_p), ("ws", c_void_p), ("ws_bytes", c_size_t),
        ("planes", c_int),
    ]


class VitEmbedArgs(C.Structure):
    _fields_ = [
        ("pix", c_void_p), ("n", c_int), ("img", c_int), ("patch", c_int), ("kpad", c_int),
        ("patch_w", c_void_p), ("patch_b", c_void_p),
        ("cls", c_void_p), ("pos", c_void_p),
        ("ln_w", c_void_p), ("ln_b", c_void_p), ("eps", c_float),
        ("ntok", c_int), ("C", c_int),
        ("patches", c_void_p), ("pe", c_void_p),
        ("x", c_void_p), ("h", c_void_p), ("h_planes", c_void_p),
        ("ws", c_void_p), ("ws_bytes", c_size_t),
    ]


class MixtralCfg(C.Structure):
    _fields_ = [
        ("hidden", c_int), ("n_layers", c_int), ("n_q_heads", c_int), ("n_kv_heads", c_int),
        ("head_dim", c_int), ("inter", c_int), ("n_experts", c_int), ("top_k", c_int), ("vocab", c_int),
        ("rms_eps", c_float), ("max_ctx", c_int), ("max_prefill", c_int), ("max_new", c_int),
        ("tp_rank", c_int), ("tp_world", c_int), ("nsplit", c_int), ("logit_rows", c_int),
        ("vocab_lo", c_int), ("vocab_n", c_int), ("max_seqs", c_int),
    ]


class MixtralLayer(C.Structure):
    _fields_ = [
        ("attn_norm", c_void_p), ("wqkv", c_void_p), ("wo", c_void_p), ("ffn_norm", c_void_p),
        ("wrouter", c_void_p), ("w1", c_void_p), ("w3", c_void_p), ("w2", c_void_p),
    ]


ALLREDUCE_FN = C.CFUNCTYPE(c_int, c_void_p, c_void_p, c_long, c_void_p)

# every symbol include/vita_hip.h declares: (restype, argtypes)
SIGNATURES = {
    "vh_version": (c_int, []),
    "vh_last_error": (C.c_char_p, []),
    "vh_tune": (c_int, [C.c_char_p, c_int]),
    "vh_gemm": (c_int, [C.POINTER(GemmArgs), c_void_p]),
    "vh_gemm_ln": (c_int, [C.POINTER(GemmArgs), c_void_p, c_void_p, c_float, c_void_p, c_long, c_void_p]),
    "vh_gemm_ps": (c_int, [C.POINTER(GemmPsArgs), c_void_p]),
    "vh_split_planes": (c_int, [c_void_p, c_long, c_void_p, c_void_p, c_long, c_int, c_int, c_void_p]),
    "vh_attention": (c_int, [C.POINTER(AttnArgs), c_void_p]),
    "vh_encoder_layer": (c_int, [C.POINTER(EncoderLayerArgs), c_void_p]),
    "vh_layernorm": (c_int, [c_void_p, c_long, c_void_p, c_long, c_void_p, c_void_p, c_int, c_int, c_float, c_int,
                             c_float, c_void_p]),
    "vh_rmsnorm": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_void_p]),
    "vh_add": (c_int, [c_void_p, c_void_p, c_long, c_void_p]),
    "vh_cast_bf16_f32": (c_int, [c_void_p, c_void_p, c_long, c_void_p]),
    "vh_fill_hash_bf16": (c_int, [c_void_p, c_long, c_long, c_long, c_long, c_long, C.c_uint64, c_void_p]),
    "vh_vit_embed": (c_int, [c_void_p, c_void_p]),
    "vh_vit_patchify": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "vh_vit_assemble": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "vh_vit_pixel_shuffle": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p]),
    "vh_audio_conv1": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                               c_void_p]),
    "vh_embed_splice": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "vh_router_top2": (c_int, [c_void_p, c_long, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "vh_moe_decode": (c_int, [c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                              c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "vh_rope_kv_append": (c_int, [c_void_p, c_long, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                  c_int, c_void_p, c_void_p]),
    "vh_attn_decode": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p,
                               c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "vh_lmhead_argmax": (c_int, [c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                 c_void_p, c_int, c_void_p]),
    "vh_mixtral_workspace_bytes": (c_size_t, [C.POINTER(MixtralCfg)]),
    "vh_mixtral_create": (c_void_p, [C.POINTER(MixtralCfg), C.POINTER(MixtralLayer), c_void_p, c_void_p, c_void_p,
                                     c_void_p, c_void_p, c_void_p, c_size_t]),
    "vh_mixtral_destroy": (None, [c_void_p]),
    "vh_mixtral_set_allreduce": (c_int, [c_void_p, ALLREDUCE_FN, c_void_p]),
    "vh_rccl_unique_id": (c_int, [c_void_p]),
    "vh_mixtral_init_rccl": (c_int, [c_void_p, c_void_p]),
    "vh_comm_create": (c_void_p, [c_int, c_int, c_size_t, c_void_p]),
    "vh_comm_create_loopback": (c_void_p, [c_int, c_int, c_size_t]),
    "vh_comm_connect": (c_int, [c_void_p, c_void_p]),
    "vh_comm_capacity": (c_size_t, [c_void_p]),
    "vh_comm_allreduce": (c_int, [c_void_p, c_void_p, c_long, c_void_p]),
    "vh_comm_status": (c_int, [c_void_p]),
    "vh_comm_destroy": (None, [c_void_p]),
    "vh_comm_last_error": (C.c_char_p, []),
    "vh_comm_debug_set_calls": (c_int, [c_void_p, C.c_uint64]),
    "vh_comm_is_fine_grained": (c_int, [c_void_p]),
    "vh_mixtral_use_comm": (c_int, [c_void_p, c_void_p]),
    "vh_mixtral_cancel_rccl": (c_int, [c_void_p]),
    "vh_mixtral_route_debug": (c_int, [c_void_p, c_void_p]),
    "vh_mixtral_prefill": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "vh_mixtral_decode": (c_int, [c_void_p, c_int, c_void_p]),
    "vh_mixtral_tokens": (c_void_p, [c_void_p]),
    "vh_mixtral_counters": (c_void_p, [c_void_p]),
    "vh_mixtral_logits": (c_void_p, [c_void_p]),
    "vh_mixtral_reset": (c_int, [c_void_p, c_void_p]),
    "vh_mixtral_decode_schedule": (c_int, [c_void_p]),
    "vh_mixtral_seq_alloc": (c_int, [c_void_p]),
    "vh_mixtral_seq_free": (c_int, [c_void_p, c_int]),
    "vh_mixtral_seq_prefill": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p]),
    "vh_mixtral_seq_decode": (c_int, [c_void_p, C.POINTER(c_int), c_int, c_void_p]),
    "vh_mixtral_pages_free": (c_int, [c_void_p]),
    "vh_mixtral_seq_pos": (c_int, [c_void_p, c_int]),
    "vh_mixtral_seq_tokens": (c_void_p, [c_void_p, c_int]),
    "vh_mixtral_seq_counters": (c_void_p, [c_void_p, c_int]),
    "vh_mixtral_seq_table": (c_int, [c_void_p, c_int, C.POINTER(c_int), c_int]),
    "vh_mixtral_profile": (c_int, [c_void_p, c_int, c_int]),
    "vh_mixtral_profile_read": (c_int, [c_void_p, C.POINTER(C.c_double), C.POINTER(c_int)]),
}

_lib = None


class VitaHipError(RuntimeError):
    pass


def load():
    """Load the shared library and bind every declared symbol (raises if any is missing)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise VitaHipError(
            f"{LIB_PATH} not found: the HIP extension is not built (run `python __graft_entry__.py`). "
            "There is no CPU fallback for the vita_amd hot path.")
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name, None)
        if fn is None:
            raise VitaHipError(f"libvita_hip.so does not export {name}")
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def tune(key, value):
    check(load().vh_tune(key.encode(), int(value)), f"vh_tune({key})")


def check(rc, what=""):
    if rc != 0:
        msg = load().vh_last_error()
        raise VitaHipError(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")
