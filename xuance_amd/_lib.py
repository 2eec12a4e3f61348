"""ctypes binding of libxrl_hip.so (include/xrl_hip.h).  The product path has NO CPU fallback: if the shared
library is missing or a call fails, an exception is raised."""
import ctypes as C
import os

from .build import LIB_PATH

c_void_p, c_int, c_int32, c_int64, c_float, c_double = C.c_void_p, C.c_int, C.c_int32, C.c_int64, C.c_float, C.c_double

ACT = {None: 0, "none": 0, "relu": 1, "leaky_relu": 2, "tanh": 3, "sigmoid": 4}


class XrlError(RuntimeError):
    pass


class Field(C.Structure):
    _fields_ = [("dst", c_void_p), ("src", c_void_p), ("row_bytes", c_int32), ("flags", c_int32)]


class Gemm(C.Structure):
    _fields_ = [("A", c_void_p), ("B", c_void_p), ("C", c_void_p), ("bias", c_void_p), ("dbias", c_void_p),
                ("aux", c_void_p), ("M", c_int32), ("N", c_int32), ("K", c_int32), ("lda", c_int32), ("ldb", c_int32),
                ("ldc", c_int32), ("ldaux", c_int32), ("act", c_int32), ("pad", c_int32)]


class Conv(C.Structure):
    _fields_ = [("img", c_void_p), ("w", c_void_p), ("bias", c_void_p), ("mask", c_void_p), ("out", c_void_p), ("dy", c_void_p),
                ("dbias", c_void_p), ("B", c_int32), ("IH", c_int32), ("IW", c_int32), ("C", c_int32), ("Th", c_int32),
                ("Tw", c_int32), ("nh", c_int32), ("nw", c_int32), ("sh", c_int32), ("off_h", c_int32), ("off_w", c_int32),
                ("so", c_int32), ("ph", c_int32), ("pw", c_int32), ("OHt", c_int32), ("OWt", c_int32), ("N", c_int32),
                ("act", c_int32), ("img_u8", c_int32), ("pad", c_int32)]


class DqnHeadTd(C.Structure):
    _fields_ = [("h_eval", c_void_p), ("h_target", c_void_p), ("w_eval", c_void_p), ("b_eval", c_void_p), ("w_target", c_void_p),
                ("b_target", c_void_p), ("actions", c_void_p), ("rewards", c_void_p), ("terminals", c_void_p), ("q_eval", c_void_p),
                ("q_target", c_void_p), ("d_q", c_void_p), ("d_h", c_void_p), ("diag", c_void_p), ("partials", c_void_p),
                ("M", c_int32), ("A", c_int32), ("H", c_int32), ("ld_h", c_int32), ("ld_q", c_int32), ("double_q", c_int32),
                ("act", c_int32), ("pad", c_int32), ("gamma", c_float), ("huber_delta", c_float)]


class DqnTailTd(C.Structure):
    _fields_ = [(k, c_void_p) for k in ("y_eval", "y_target", "feat_eval", "feat_target", "arg", "w1_eval", "b1_eval", "w1_target",
                                        "b1_target", "w2_eval", "b2_eval", "w2_target", "b2_target", "actions", "rewards", "terminals",
                                        "q_eval", "q_target", "d_q", "h_eval", "d_h", "d_feat", "dy", "diag", "partials")] + \
               [(k, c_int32) for k in ("M", "A", "H", "F", "P", "ld_h", "ld_q", "ld_f", "double_q", "act")] + \
               [("gamma", c_float), ("huber_delta", c_float), ("slabs", c_void_p)] + \
               [(k, C.c_int64) for k in ("slab_stride", "off_w1", "off_b1", "off_w2", "off_b2")]


class DqnActTail(C.Structure):
    _fields_ = [(k, c_void_p) for k in ("y", "w1", "b1", "w2", "b2", "eps_dev", "action", "action_f", "q", "feat", "step_dev")] + \
               [("seed", C.c_uint64), ("step", C.c_uint32)] + \
               [(k, c_int32) for k in ("n", "A", "H", "F", "P", "ld_q", "ld_f", "act")] + [("eps", c_float)] + \
               [("eps_sched", c_int32), ("eps_n", c_int32), ("eps_kstar", C.c_uint32), ("pad1", C.c_uint32),
                ("eps_start", C.c_double), ("eps_delta", C.c_double)]


class ImageJob(C.Structure):
    _fields_ = [("src", c_void_p), ("map", c_void_p), ("dst", c_void_p), ("n", C.c_int64)]


class PpoLoss(C.Structure):
    _fields_ = [("out", c_void_p), ("value", c_void_p), ("actions", c_void_p), ("adv", c_void_p), ("stats", c_void_p),
                ("returns", c_void_p), ("old_logp", c_void_p), ("log_std", c_void_p), ("d_out", c_void_p),
                ("d_value", c_void_p), ("d_log_std", c_void_p), ("diag", c_void_p), ("partials", c_void_p),
                ("M", c_int32), ("A", c_int32), ("ld_out", c_int32), ("ld_v", c_int32), ("out_act", c_int32),
                ("n_split", c_int32), ("slab_stride", c_int64), ("clip_range", c_float), ("vf_coef", c_float),
                ("ent_coef", c_float), ("mode", c_int32), ("pad_mode", c_int32), ("old_a", c_void_p), ("old_b", c_void_p),
                ("kl_coef", c_void_p)]


class AdamState(C.Structure):
    _fields_ = [("step", c_int32), ("sched_steps", c_int32), ("total_iters", c_int32), ("ticket", c_int32),
                ("base_lr", c_double), ("end_factor", c_double), ("beta1", c_double), ("beta2", c_double),
                ("eps", c_double), ("weight_decay", c_double), ("last_lr", c_double), ("last_grad_norm", c_double)]


class Rms(C.Structure):
    _fields_ = [("x", c_void_p), ("mean", c_void_p), ("var", c_void_p), ("count", c_void_p), ("out0", c_void_p),
                ("out1", c_void_p), ("n", c_int), ("D", c_int), ("ld_x", c_int), ("ld0", c_int), ("ld1", c_int),
                ("update", c_int), ("normalize", c_int), ("range", c_float)]


class Sample(C.Structure):
    _fields_ = [("heads", c_void_p), ("log_std", c_void_p), ("noise", c_void_p), ("act_out", c_void_p),
                ("val_out", c_void_p), ("logp_out", c_void_p), ("env_action", c_void_p), ("env_action_f", c_void_p),
                ("bootv_prev", c_void_p), ("n", c_int), ("A", c_int), ("ld", c_int), ("gaussian", c_int),
                ("seed", C.c_uint64), ("step", C.c_uint32), ("step_dev", c_void_p)]


class CartPole(C.Structure):
    _fields_ = [("state", c_void_p), ("steps", c_void_p), ("episodes", c_void_p), ("action", c_void_p),
                ("obs", c_void_p), ("next_obs", c_void_p), ("reward", c_void_p), ("terminated", c_void_p),
                ("truncated", c_void_p), ("ep_score", c_void_p), ("stats", c_void_p), ("n", c_int),
                ("max_steps", c_int), ("seed", C.c_uint64)]


class Classic(C.Structure):
    _fields_ = [("state", c_void_p), ("steps", c_void_p), ("episodes", c_void_p), ("action", c_void_p), ("action_f", c_void_p),
                ("obs", c_void_p), ("next_obs", c_void_p), ("reward", c_void_p), ("terminated", c_void_p), ("truncated", c_void_p),
                ("ep_score", c_void_p), ("stats", c_void_p), ("n", c_int32), ("kind", c_int32), ("max_steps", c_int32),
                ("pad", c_int32), ("seed", C.c_uint64)]


class ActTail(C.Structure):
    _fields_ = [("hb", c_void_p), ("w_actor", c_void_p), ("b_actor", c_void_p), ("w_critic", c_void_p), ("b_critic", c_void_p),
                ("heads", c_void_p), ("ldh", c_int32), ("K", c_int32), ("a_off", c_int32), ("c_off", c_int32), ("ldw_a", c_int32),
                ("ldw_c", c_int32), ("boot_rows", c_int32), ("boot_actor", c_int32), ("env_kind", c_int32), ("act_actor", c_int32),
                ("sample", Sample), ("classic", Classic), ("cartpole", CartPole)]


class PostStep(C.Structure):
    _fields_ = [("reward", c_void_p), ("terminated", c_void_p), ("truncated", c_void_p), ("next_obs", c_void_p),
                ("obs_mean", c_void_p), ("obs_var", c_void_p), ("next_obs_norm", c_void_p), ("rew_out", c_void_p),
                ("term_out", c_void_p), ("seg_out", c_void_p), ("ret_track", c_void_p), ("ret_mean", c_void_p),
                ("ret_var", c_void_p), ("ret_count", c_void_p), ("n", c_int), ("D", c_int), ("ld_next", c_int),
                ("use_obsnorm", c_int), ("use_rewnorm", c_int), ("last_step", c_int), ("obs_range", c_float),
                ("rew_range", c_float), ("gamma", c_float), ("pg_bootv", c_void_p)]


class PpoActTail(C.Structure):
    """xrl_ppo_act_tail_t: split-K epilogue + heads + sampling (+ the previous step's bookkeeping, + the observation copy) in one launch."""
    _fields_ = [("ws", c_void_p), ("bias", c_void_p), ("ks", c_int32), ("M", c_int32), ("H", c_int32), ("act", c_int32),
                ("w_actor", c_void_p), ("b_actor", c_void_p), ("w_critic", c_void_p), ("b_critic", c_void_p),
                ("noise", c_void_p), ("act_out", c_void_p), ("val_out", c_void_p), ("logp_out", c_void_p), ("env_action", c_void_p),
                ("bootv_prev", c_void_p), ("n", c_int32), ("A", c_int32), ("seed", C.c_uint64), ("step", C.c_uint32), ("pad0", C.c_uint32),
                ("step_dev", c_void_p), ("heads_out", c_void_p), ("post", PostStep), ("post_n", c_int32), ("pad1", c_int32),
                ("copy_src", c_void_p), ("copy_dst", c_void_p), ("copy_bytes", c_int64)]


class EGreedy(C.Structure):
    _fields_ = [("q", c_void_p), ("uniforms", c_void_p), ("randoms", c_void_p), ("eps_dev", c_void_p),
                ("action", c_void_p), ("action_f", c_void_p), ("n", c_int), ("A", c_int), ("ld", c_int), ("eps", c_float),
                ("seed", C.c_uint64), ("step", C.c_uint32), ("step_dev", c_void_p)]


class DqnTd(C.Structure):
    _fields_ = [("q_eval", c_void_p), ("q_next", c_void_p), ("q_next_eval", c_void_p), ("actions", c_void_p),
                ("rewards", c_void_p), ("terminals", c_void_p), ("d_q", c_void_p), ("diag", c_void_p),
                ("partials", c_void_p), ("M", c_int32), ("A", c_int32), ("ld", c_int32), ("n_split", c_int32),
                ("gamma", c_float), ("dueling", c_int32), ("huber_delta", c_float), ("pad", c_int32)]


class Qmix(C.Structure):
    _fields_ = [("q_eval", c_void_p), ("q_next_eval", c_void_p), ("q_next", c_void_p), ("actions", c_void_p),
                ("avail_next", c_void_p), ("agent_mask", c_void_p), ("rewards", c_void_p), ("terminals", c_void_p),
                ("e_b1", c_void_p), ("e_raw", c_void_p), ("t_b1", c_void_p), ("t_raw", c_void_p), ("d_q", c_void_p),
                ("d_e_b1", c_void_p), ("d_e_raw", c_void_p), ("diag", c_void_p), ("partials", c_void_p),
                ("B", c_int32), ("N", c_int32), ("A", c_int32), ("H", c_int32), ("ldq", c_int32), ("ld_e1", c_int32),
                ("ld_e2", c_int32), ("ld_t1", c_int32), ("ld_t2", c_int32), ("double_q", c_int32),
                ("gamma", c_float), ("mixer", c_int32), ("filled", c_void_p)]


class EpisodeField(C.Structure):
    _fields_ = [("a", c_void_p), ("b", c_void_p), ("c", c_void_p), ("row_bytes", c_int32), ("slots", c_int32),
                ("flags", c_int32), ("pad", c_int32), ("d", c_void_p)]


class GruFwd(C.Structure):
    _fields_ = [("gi", c_void_p), ("w_hh", c_void_p), ("b_hh", c_void_p), ("h0", c_void_p), ("reset", c_void_p),
                ("hs", c_void_p), ("gates", c_void_p), ("h_last", c_void_p),
                ("R", c_int32), ("T1", c_int32), ("H", c_int32), ("ld_gi", c_int32),
                ("gi2", c_void_p), ("w_hh2", c_void_p), ("b_hh2", c_void_p), ("hs2", c_void_p)]


class GruBwd(C.Structure):
    _fields_ = [("d_hs", c_void_p), ("hs", c_void_p), ("gates", c_void_p), ("w_hh", c_void_p), ("d_gi", c_void_p),
                ("d_gh", c_void_p), ("d_h0", c_void_p),
                ("R", c_int32), ("T1", c_int32), ("H", c_int32), ("ld_dhs", c_int32), ("ld_dgi", c_int32), ("pad", c_int32)]


class LstmFwd(C.Structure):
    _fields_ = [("gi", c_void_p), ("w_hh", c_void_p), ("b_hh", c_void_p), ("h0", c_void_p), ("c0", c_void_p), ("reset", c_void_p),
                ("hs", c_void_p), ("cs", c_void_p), ("gates", c_void_p), ("h_last", c_void_p), ("c_last", c_void_p),
                ("R", c_int32), ("T1", c_int32), ("H", c_int32), ("ld_gi", c_int32),
                ("gi2", c_void_p), ("w_hh2", c_void_p), ("b_hh2", c_void_p), ("hs2", c_void_p)]


class LstmBwd(C.Structure):
    _fields_ = [("d_hs", c_void_p), ("cs", c_void_p), ("gates", c_void_p), ("w_hh", c_void_p), ("d_gates", c_void_p),
                ("R", c_int32), ("T1", c_int32), ("H", c_int32), ("ld_dhs", c_int32), ("ld_dg", c_int32), ("pad", c_int32)]


class FusedLayer(C.Structure):
    _fields_ = [("w_off", c_int32), ("b_off", c_int32), ("K", c_int32), ("N", c_int32), ("act", c_int32),
                ("in_level", c_int32), ("in_off", c_int32), ("out_level", c_int32), ("out_off", c_int32), ("pad", c_int32)]


class MlpChainJob(C.Structure):
    _fields_ = [("x", c_void_p), ("params", c_void_p), ("layers", FusedLayer * 8), ("n_layers", c_int32), ("n_levels", c_int32),
                ("level_width", c_int32 * 6), ("out", c_void_p * 6), ("ld_out", c_int32 * 6), ("ldx", c_int32), ("M", c_int32)]


class MlpChain(C.Structure):
    _fields_ = [("job", MlpChainJob * 4), ("n_jobs", c_int32), ("tile0", c_int32 * 5), ("pad", c_int32 * 2)]


class RolloutStep(C.Structure):
    _fields_ = [("params", c_void_p), ("cache_image", c_void_p), ("frag_image", c_void_p), ("layers", FusedLayer * 8), ("n_layers", c_int32), ("n_levels", c_int32),
                ("n_head_layers", c_int32), ("pad0", c_int32), ("level_width", c_int32 * 6),
                ("obs_raw_in", c_void_p), ("obs_raw_out", c_void_p), ("xnext_in", c_void_p), ("xnext_out", c_void_p),
                ("obs_stats_in", c_void_p), ("obs_stats_out", c_void_p), ("obs_count_in", c_void_p),
                ("obs_count_out", c_void_p), ("ret_stats_in", c_void_p), ("ret_stats_out", c_void_p),
                ("ret_count_in", c_void_p), ("ret_count_out", c_void_p), ("ended_in", c_void_p), ("ended_out", c_void_p),
                ("ret_final_in", c_void_p), ("ret_final_out", c_void_p), ("ret_track", c_void_p),
                ("obs_slot", c_void_p), ("act_slot", c_void_p), ("val_slot", c_void_p), ("logp_slot", c_void_p),
                ("rew_slot", c_void_p), ("term_slot", c_void_p), ("seg_slot", c_void_p), ("bootv_prev", c_void_p),
                ("log_std", c_void_p),
                ("cp_state", c_void_p), ("cp_steps", c_void_p), ("cp_episodes", c_void_p), ("cp_score", c_void_p),
                ("cp_stats", c_void_p),
                ("n", c_int32), ("D", c_int32), ("A", c_int32), ("gaussian", c_int32), ("max_steps", c_int32),
                ("use_obsnorm", c_int32), ("use_rewnorm", c_int32), ("last_step", c_int32), ("boot_only", c_int32),
                ("role_split", c_int32), ("split_col", c_int32),
                ("obs_range", c_float), ("rew_range", c_float), ("gamma", c_float), ("pad1", c_float),
                ("seed", C.c_uint64), ("env_seed", C.c_uint64), ("step", C.c_uint32), ("step_dev", c_void_p),
                ("dbg", c_void_p)]


class SynthCtl(C.Structure):
    _fields_ = [("state", c_void_p), ("steps", c_void_p), ("action", c_void_p), ("Amat", c_void_p), ("Bmat", c_void_p),
                ("obs", c_void_p), ("next_obs", c_void_p), ("reward", c_void_p), ("terminated", c_void_p),
                ("truncated", c_void_p), ("ep_score", c_void_p), ("stats", c_void_p), ("n", c_int32), ("D", c_int32),
                ("A", c_int32), ("max_steps", c_int32), ("seed", C.c_uint64), ("step", C.c_uint32), ("step_dev", c_void_p)]


class SynthFrames(C.Structure):
    _fields_ = [("cur_obs", c_void_p), ("next_obs", c_void_p), ("action", c_void_p), ("reward", c_void_p),
                ("terminated", c_void_p), ("truncated", c_void_p), ("done", c_void_p), ("steps", c_void_p),
                ("end_step", c_void_p), ("n", c_int32), ("row_bytes", c_int32), ("A", c_int32), ("max_steps", c_int32),
                ("p_term", c_float), ("pad", c_float), ("seed", C.c_uint64), ("step", C.c_uint32), ("step_dev", c_void_p)]


class SynthMarl(C.Structure):
    _fields_ = [("buf_obs", c_void_p), ("buf_state", c_void_p), ("buf_avail", c_void_p), ("next_obs", c_void_p),
                ("next_state", c_void_p), ("next_avail", c_void_p), ("action", c_void_p), ("rewards", c_void_p),
                ("terminals", c_void_p), ("terminated", c_void_p), ("truncated", c_void_p), ("done", c_void_p),
                ("steps", c_void_p), ("end_step", c_void_p), ("n", c_int32), ("N", c_int32), ("O", c_int32), ("S", c_int32),
                ("A", c_int32), ("max_steps", c_int32), ("p_term", c_float), ("pad", c_float), ("seed", C.c_uint64),
                ("step", C.c_uint32), ("step_dev", c_void_p), ("prev_state", c_void_p), ("prev_steps", c_void_p), ("totals", c_void_p)]


class RolloutRun(C.Structure):
    """xrl_rollout_run_t (csrc/rollout_actor.hip): steps [t0, t0 + n_steps) of a CartPole rollout of the 4-128-{128-2,128-1} class."""
    _fields_ = [("params", c_void_p)] + [(k, c_int32) for k in ("w0", "b0", "w1", "b1", "wa", "ba", "wc", "bc", "act", "n", "T", "t0",
                                                                  "n_steps", "max_steps", "use_obsnorm", "use_rewnorm", "flags")] + \
               [("obs_range", c_float), ("rew_range", c_float), ("gamma", c_float), ("pad0", c_float),
                ("seed", C.c_uint64), ("env_seed", C.c_uint64), ("step", C.c_uint32), ("pad1", C.c_uint32), ("step_dev", c_void_p)] + \
               [(k, c_void_p) for k in ("obs_raw", "obs_stats", "obs_count", "ret_stats", "ret_count", "ret_track", "cp_state", "cp_steps",
                                        "cp_episodes", "cp_score", "cp_stats", "f_obs", "f_act", "f_logp", "f_rew", "f_term", "f_seg",
                                        "f_val", "bootv", "xnext", "ended", "ret_final", "xchg", "status", "dbg",
                                        "tape_next_obs", "tape_reset_obs", "tape_term", "tape_trunc", "tape_pos", "tape_u")] + \
               [("tape_rows", c_int32), ("pad2", c_int32)]


class RolloutWide(C.Structure):
    """xrl_rollout_wide_t (csrc/rollout_wide.hip): steps [t0, t0 + n_steps) of a rollout of the D-256-256-{A | 1} Gaussian class."""
    _fields_ = [("params", c_void_p)] + [(k, c_int32) for k in ("w0", "b0", "w1", "b1", "w2", "b2", "log_std_off", "act", "out_act", "D", "A", "H",
                                                                  "n", "T", "t0", "n_steps", "max_steps", "use_obsnorm", "use_rewnorm", "flags")] + \
               [("obs_range", c_float), ("rew_range", c_float), ("gamma", c_float), ("pad0", c_float),
                ("seed", C.c_uint64), ("env_seed", C.c_uint64), ("step", C.c_uint32), ("env_step", C.c_uint32),
                ("step_dev", c_void_p), ("env_step_dev", c_void_p)] + \
               [(k, c_void_p) for k in ("obs_raw", "obs_mean", "obs_var", "obs_count", "ret_mean", "ret_var", "ret_count", "ret_track", "env_state", "env_steps",
                                        "env_score", "env_stats", "Amat", "Bmat", "f_obs", "f_act", "f_logp", "f_rew", "f_term", "f_seg",
                                        "xnext", "ended", "ret_final", "raw_rew", "xchg", "status", "dbg",
                                        "tape_next_obs", "tape_reset_obs", "tape_rew", "tape_term", "tape_trunc", "tape_pos", "tape_z")] + \
               [("tape_rows", c_int32), ("pad2", c_int32)]


class PpoFused(C.Structure):
    _fields_ = [("params", c_void_p), ("params_t", c_void_p), ("cache_image", c_void_p), ("layers", FusedLayer * 8),
                ("n_layers", c_int32), ("n_levels", c_int32), ("n_head_layers", c_int32), ("pad0", c_int32),
                ("level_width", c_int32 * 6),
                ("f_obs", c_void_p), ("f_act", c_void_p), ("f_ret", c_void_p), ("f_adv", c_void_p), ("f_logp", c_void_p),
                ("idx", c_void_p), ("stats", c_void_p), ("slabs", c_void_p), ("partials", c_void_p), ("diag", c_void_p),
                ("slab_stride", c_int64), ("M", c_int32), ("n_envs", c_int32), ("T", c_int32), ("D", c_int32),
                ("A", c_int32), ("l0_fold_off", c_int32), ("clip_range", c_float), ("vf_coef", c_float), ("ent_coef", c_float),
                ("pad2", c_float), ("dbg", c_void_p), ("frag_image", c_void_p), ("f_rows", c_void_p), ("f_packed", c_void_p),
                ("dist", c_int32), ("out_act", c_int32), ("log_std_off", c_int32), ("pad3", c_int32), ("frag16", c_void_p),
                ("fwd_out", c_void_p), ("fwd_ld", c_int32), ("pad4", c_int32)]


class WideBranch(C.Structure):
    _fields_ = [("w0", c_int32), ("b0", c_int32), ("w1", c_int32), ("b1", c_int32), ("w2", c_int32), ("b2", c_int32)]


class PpoWide(C.Structure):
    _fields_ = [("params", c_void_p), ("frag", c_void_p), ("br", WideBranch * 2), ("log_std_off", c_int32),
                ("D", c_int32), ("A", c_int32), ("H", c_int32), ("act", c_int32), ("out_act", c_int32),
                ("M", c_int32), ("dbg_role", c_int32),
                ("obs", c_void_p), ("actions", c_void_p), ("ret", c_void_p), ("adv", c_void_p), ("old_logp", c_void_p),
                ("stats", c_void_p), ("slabs", c_void_p), ("slab_stride", c_int64), ("partials", c_void_p), ("diag", c_void_p),
                ("heads", c_void_p), ("clip_range", c_float), ("vf_coef", c_float), ("ent_coef", c_float), ("pad0", c_float),
                ("dbg", c_void_p), ("rows_g2", c_void_p), ("rows_h1", c_void_p), ("rows_ld", c_int64)]


class WideAct(C.Structure):
    _fields_ = [("params", c_void_p), ("frag", c_void_p), ("br", WideBranch * 2), ("log_std_off", c_int32),
                ("D", c_int32), ("A", c_int32), ("H", c_int32), ("act", c_int32), ("out_act", c_int32),
                ("n", c_int32), ("flags", c_int32), ("x", c_void_p), ("act_out", c_void_p), ("env_action_f", c_void_p),
                ("logp_out", c_void_p), ("val_out", c_void_p), ("bootv_prev", c_void_p), ("seed", C.c_uint64),
                ("step", C.c_uint32), ("pad0", C.c_uint32), ("step_dev", c_void_p),
                ("raw", c_void_p), ("mean_in", c_void_p), ("var_in", c_void_p), ("count_in", c_void_p),
                ("mean_out", c_void_p), ("var_out", c_void_p), ("count_out", c_void_p), ("obs_slot", c_void_p),
                ("update", c_int32), ("normalize", c_int32), ("range", c_float), ("pad1", c_float),
                ("xchg", c_void_p), ("xcnt", c_void_p), ("next_raw", c_void_p), ("post", PostStep), ("has_post", c_int32),
                ("pad2", c_int32), ("dbg", c_void_p)]


class QfImage(C.Structure):
    _fields_ = [("w", c_int32 * 4), ("b", c_int32 * 4), ("ldw", c_int32 * 4), ("mw", c_int32 * 5), ("mb", c_int32 * 5),
                ("mldw", c_int32 * 5), ("agent_floats", c_int32), ("mixer_floats", c_int32)]


class QmixFused(C.Structure):
    _fields_ = [("img_eval", c_void_p), ("img_target", c_void_p), ("n_layers", c_int32), ("act", c_int32), ("dims", c_int32 * 5),
                ("products", c_int32), ("w_off", c_int64 * 4), ("b_off", c_int64 * 4), ("mix_off", c_int64 * 10),
                ("N", c_int32), ("A", c_int32), ("S", c_int32), ("H", c_int32), ("HH", c_int32),
                ("B", c_int32), ("items_per_wg", c_int32), ("double_q", c_int32),
                ("obs", c_void_p), ("obs_next", c_void_p), ("state", c_void_p), ("state_next", c_void_p), ("actions", c_void_p),
                ("rewards", c_void_p), ("terminals", c_void_p), ("agent_mask", c_void_p), ("avail_next", c_void_p),
                ("slabs", c_void_p), ("slab_stride", c_int64), ("partials", c_void_p), ("diag", c_void_p),
                ("gamma", c_float), ("pad1", c_float), ("dbg", c_void_p),
                ("ring_n_envs", c_int32), ("ring_n_size", c_int32), ("size_dev", c_void_p), ("counter_dev", c_void_p),
                ("idx_out", c_void_p), ("draw_seed", C.c_uint64), ("draw_counter", C.c_uint32), ("pad3", C.c_uint32)]


class QmixPhase(C.Structure):
    """xrl_qmix_phase_t: the optimiser's side of a whole update phase done by one launch (xrl_qmix_fused_phase)."""
    _fields_ = [("n_updates", c_int32), ("sync_every", c_int32), ("params", c_void_p), ("grad", c_void_p), ("m", c_void_p), ("v", c_void_p),
                ("P", c_int64), ("state", c_void_p), ("map", c_void_p), ("target", c_void_p), ("act_image", c_void_p), ("act_map", c_void_p),
                ("phase_partials", c_void_p), ("epoch_sums", c_void_p), ("sumsq_part", c_void_p), ("scalars", c_void_p), ("tick", c_void_p),
                ("tick_inc", c_int32), ("pad", c_int32), ("sync", c_void_p)]


QF_PHASE_SYNC_WORDS = 256


class QaImage(C.Structure):
    _fields_ = [("w", c_int32 * 8), ("b", c_int32 * 8), ("ldw", c_int32 * 8), ("image_floats", c_int32), ("lds_bytes", c_int32),
                ("interleaved", c_int32), ("pad", c_int32)]


class MarlActGru(C.Structure):
    _fields_ = [("image", c_void_p), ("obs", c_void_p), ("h", c_void_p), ("reset", c_void_p), ("q", c_void_p),
                ("R", c_int32), ("rows_per_wg", c_int32), ("O", c_int32), ("H", c_int32), ("ldq", c_int32), ("act", c_int32),
                ("n_pre", c_int32), ("n_post", c_int32), ("pre", c_int32 * 3), ("post", c_int32 * 3),
                ("action", c_void_p), ("action_f", c_void_p), ("avail", c_void_p), ("eps_dev", c_void_p), ("step_dev", c_void_p),
                ("seed", C.c_uint64), ("step", C.c_uint32), ("eps", c_float), ("lds_staged", c_int32), ("pad", c_int32)]


class Exchange(C.Structure):
    _fields_ = [("base", c_void_p * 8), ("stride4", c_int64), ("world", c_int32), ("rank", c_int32),
                ("max_spins", c_int32), ("pad", c_int32), ("inv_world", c_float), ("pad2", c_float)]


class MarlGate(C.Structure):
    _fields_ = [("totals", c_void_p), ("base", c_void_p), ("call", c_void_p), ("snap", c_void_p), ("active", c_void_p),
                ("e_state", c_void_p), ("eps_dev", c_void_p), ("active_f", c_void_p), ("active_i", c_void_p),
                ("host_flags", c_void_p), ("seq", c_void_p),
                ("start_greedy", C.c_double), ("end_greedy", C.c_double), ("delta_greedy", C.c_double),
                ("ring", c_int32), ("reset_rule", c_int32), ("done", c_void_p), ("reset_rows", c_void_p), ("counters", c_void_p),
                ("n_envs", c_int32), ("n_agents", c_int32), ("ptr_size", c_void_p), ("buffer_size", c_int32), ("pad2", c_int32),
                ("end_step", c_void_p), ("next_state", c_void_p), ("stored_state", c_void_p), ("state_dim", c_int32), ("pad3", c_int32)]


class Mirrors(C.Structure):
    _fields_ = [("map", c_void_p * 4), ("dst", c_void_p * 4), ("n", c_int32), ("target_every", c_int32), ("target", c_void_p),
                ("target_image", c_void_p), ("fold_off", c_int64), ("fold_len", c_int32), ("split_plane", c_int32), ("tick", c_void_p),
                ("part", c_void_p), ("part_out", c_void_p), ("tick_inc", c_int32), ("part_rows", c_int32),
                ("alt_lo", c_int64 * 2), ("alt_hi", c_int64 * 2), ("alt_split", c_int32), ("pad2", c_int32)]


class OptChain(C.Structure):
    """xrl_opt_chain_t: the optimiser step of the previous minibatch, done by the next minibatch launch (xrl_ppo_trunk_chained)."""
    _fields_ = [("slabs", c_void_p), ("slab_stride", c_int64), ("params", c_void_p), ("grad", c_void_p), ("m", c_void_p),
                ("v", c_void_p), ("P", c_int64), ("state", c_void_p), ("sumsq_part", c_void_p), ("max_norm", C.c_double),
                ("sync", c_void_p), ("n_split", c_int32), ("n_part", c_int32), ("mirrors", Mirrors)]


CHAIN_MAX_WGS = 512
CHAIN_SYNC_WORDS = 4 + 2 * CHAIN_MAX_WGS


class MarlAct(C.Structure):
    _fields_ = [("q", c_void_p), ("avail", c_void_p), ("eps_dev", c_void_p), ("coin", c_void_p), ("uniforms", c_void_p),
                ("action", c_void_p), ("action_f", c_void_p), ("R", c_int32), ("A", c_int32), ("ld", c_int32),
                ("eps", c_float), ("seed", C.c_uint64), ("step", C.c_uint32), ("step_dev", c_void_p)]


_SIGS = {
    "xrl_marl_select_actions": [C.POINTER(MarlAct), c_void_p],
    "xrl_ppo_fused_minibatch": [C.POINTER(PpoFused), c_void_p],
    "xrl_ppo_trunk_chained": [C.POINTER(PpoFused), C.POINTER(OptChain), c_void_p],
    "xrl_ppo_trunk_chain_fits": [c_int32, c_int32, c_int64],
    "xrl_transpose_mid": [C.POINTER(PpoFused), c_void_p, c_void_p],
    "xrl_pack_mid_frags": [C.POINTER(PpoFused), c_void_p, c_int64, c_void_p],
    "xrl_pack_mid_frags16": [C.POINTER(PpoFused), c_void_p, c_int64, c_void_p],
    "xrl_trunk_forward16": [C.POINTER(PpoFused), c_void_p, c_void_p],
    "xrl_set_split_product_tr": [c_int32],
    "xrl_set_rollout_split_products": [c_int32],
    "xrl_set_split_product_ksplit": [c_int32],
    "xrl_ppo_wide_minibatch": [C.POINTER(PpoWide), c_void_p],
    "xrl_ppo_wide_pack": [C.POINTER(PpoWide), c_void_p, c_void_p],
    "xrl_wide_dw1": [C.POINTER(PpoWide), C.POINTER(c_int32), c_void_p],
    "xrl_wide_act_step": [C.POINTER(WideAct), c_void_p],
    "xrl_gather_rows": [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p],
    "xrl_pack_transitions": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p],
    "xrl_init": [],
    "xrl_rollout_step_cartpole": [C.POINTER(RolloutStep), c_void_p],
    "xrl_pack_rollout_cache": [C.POINTER(RolloutStep), c_void_p, c_int64, c_void_p],
    "xrl_pack_rollout_cache2": [C.POINTER(RolloutStep), c_void_p, c_int64, c_void_p, c_void_p],
    "xrl_dqn_td": [C.POINTER(DqnTd), c_void_p],
    "xrl_qmix_mix_td": [C.POINTER(Qmix), c_void_p],
    "xrl_episode_store_step": [C.POINTER(EpisodeField), c_int, c_void_p, c_int, c_void_p],
    "xrl_episode_finish": [C.POINTER(EpisodeField), c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p],
    "xrl_episode_store_finish": [C.POINTER(EpisodeField), c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p],
    "xrl_episode_store_finish_gate": [C.POINTER(EpisodeField), c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                      C.POINTER(MarlGate), c_void_p, c_void_p],
    "xrl_episode_finish_gated": [C.POINTER(EpisodeField), c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p],
    "xrl_episode_gather": [C.POINTER(EpisodeField), c_int, c_void_p, c_int, c_void_p],
    "xrl_episode_gather_sampled": [C.POINTER(EpisodeField), c_int, c_void_p, c_int, c_int, c_void_p, C.c_uint64, C.c_uint32, c_void_p,
                                   c_void_p],
    "xrl_marl_loop_gate": [C.POINTER(MarlGate), c_void_p],
    "xrl_rollout_cartpole_max_envs": [],
    "xrl_rollout_wide_max_envs": [],
    "xrl_marl_stored_state": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p],
    "xrl_ppokl_adapt": [c_void_p, c_int, c_double, c_void_p, c_double, c_void_p, c_void_p],
    "xrl_set_conv_dw_mixed": [c_int],
    "xrl_qmix_fused_update": [C.POINTER(QmixFused), c_void_p],
    "xrl_qmix_fused_phase": [C.POINTER(QmixFused), C.POINTER(QmixPhase), c_void_p],
    "xrl_qmix_fused_phase_fits": [c_int32, c_int32, c_int64],
    "xrl_qmix_fused_lds_bytes": [C.POINTER(QmixFused)],
    "xrl_qmix_fused_layout": [C.POINTER(QmixFused), C.POINTER(QfImage)],
    "xrl_marl_act_gru": [C.POINTER(MarlActGru), c_void_p],
    "xrl_mlp_chain_fwd": [C.POINTER(MlpChain), c_void_p],
    "xrl_mlp_chain_lds_bytes": [C.POINTER(MlpChain)],
    "xrl_debug_mlp_chain_stamps": [c_void_p],
    "xrl_act_tail": [C.POINTER(ActTail), c_void_p],
    "xrl_debug_act_tail_stamps": [c_void_p],
    "xrl_post_norm": [C.POINTER(PostStep), C.POINTER(Rms), c_void_p],
    "xrl_marl_act_gru_layout": [C.POINTER(MarlActGru), C.POINTER(QaImage)],
    "xrl_debug_act_gru_stamps": [C.c_void_p],
    "xrl_host_device_pointer": [c_void_p, C.POINTER(c_void_p)],
    "xrl_per_store": [c_void_p, c_void_p, c_void_p, c_int, C.c_double, c_int, c_int, c_void_p],
    "xrl_per_sample": [c_void_p, c_void_p, c_void_p, c_int, C.c_double, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                       c_void_p, c_void_p],
    "xrl_per_update_priorities": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, C.c_double, c_int, c_int, c_int, c_void_p],
    "xrl_lstm_forward": [C.POINTER(LstmFwd), c_void_p],
    "xrl_lstm_backward": [C.POINTER(LstmBwd), c_void_p],
    "xrl_synth_marl_step": [C.POINTER(SynthMarl), c_int, c_void_p],
    "xrl_synth_frames_step": [C.POINTER(SynthFrames), c_int, c_void_p],
    "xrl_gru_forward": [C.POINTER(GruFwd), c_void_p],
    "xrl_gru_backward": [C.POINTER(GruBwd), c_void_p],
    "xrl_sync_target": [c_void_p, c_void_p, c_int64, c_void_p, c_int, c_void_p],
    "xrl_obs_normalize": [C.POINTER(Rms), c_void_p],
    "xrl_policy_sample": [C.POINTER(Sample), c_void_p],
    "xrl_cartpole_step": [C.POINTER(CartPole), c_int, c_void_p],
    "xrl_classic_step": [C.POINTER(Classic), c_int, c_void_p],
    "xrl_rollout_poststep": [C.POINTER(PostStep), c_void_p],
    "xrl_egreedy": [C.POINTER(EGreedy), c_void_p],
    "xrl_counter_add": [c_void_p, C.c_uint32, c_void_p],
    "xrl_set_fast_kernels": [C.c_int],
    "xrl_synth_control_step": [C.POINTER(SynthCtl), c_int, c_void_p],
    "xrl_im2col_nhwc": [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "xrl_col2im_nhwc": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "xrl_maxpool_hw_fwd": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p],
    "xrl_maxpool_hw_bwd": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p],
    "xrl_reduce_adam_fits": [c_int64, c_int],
    "xrl_dqn_head_td": [c_void_p, c_void_p],
    "xrl_dqn_tail_td": [c_void_p, c_void_p],
    "xrl_dqn_act_tail": [c_void_p, c_void_p],
    "xrl_conv_fwd": [c_void_p, c_int, c_int, c_void_p],
    "xrl_conv_fwd_probe": [c_void_p, c_int, c_int, c_void_p, c_void_p],
    "xrl_conv_bwd_weight": [c_void_p, c_int, c_int, C.c_int64, c_void_p],
    "xrl_conv_bwd_weight_probe": [c_void_p, c_int, c_int, C.c_int64, c_void_p, c_void_p],
    "xrl_gather_images": [c_void_p, c_int, c_void_p],
    "xrl_flatten_chw_fwd": [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p],
    "xrl_flatten_chw_bwd": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p],
    "xrl_rollout_cartpole_run": [C.POINTER(RolloutRun), c_void_p],
    "xrl_rollout_wide_run": [C.POINTER(RolloutWide), c_void_p],
    "xrl_copy_column": [c_void_p, C.c_int, c_void_p, c_int64, c_void_p],
    "xrl_rollout_cartpole_values": [C.POINTER(RolloutRun), c_void_p],
    "xrl_sample_replay_indices": [c_void_p, c_int, c_int, c_int, c_void_p, C.c_uint64, C.c_uint32, c_void_p, c_void_p],
    "xrl_random_permutation": [c_void_p, c_int, c_int64, c_int64, C.c_uint64, C.c_uint32, c_void_p, c_void_p],
    "xrl_device_info": [C.POINTER(c_int), C.POINTER(c_int), C.c_char_p, c_int],
    "xrl_soa_store_step": [C.POINTER(Field), c_int, c_int, c_int, c_void_p],
    "xrl_soa_store_step_sized": [C.POINTER(Field), c_int, c_int, c_int, c_void_p, c_int32, c_void_p],
    "xrl_soa_store_step_ring": [C.POINTER(Field), c_int, c_int, c_int, C.c_int64, C.c_int64, c_void_p, c_int32, c_void_p, c_void_p],
    "xrl_gae_scan": [c_void_p] * 7 + [c_int, c_int, c_double, c_double, c_int, c_void_p],
    "xrl_adv_stats": [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p],
    "xrl_soa_gather": [C.POINTER(Field), c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p],
    "xrl_soa_gather_sampled": [C.POINTER(Field), c_int, c_void_p, c_int, c_int, c_int, c_void_p, C.c_uint64, C.c_uint32, c_void_p,
                               c_void_p],
    "xrl_linear_fwd": [C.POINTER(Gemm), c_int, c_void_p],
    "xrl_linear_fwd_partials": [C.POINTER(Gemm), c_int, c_void_p],
    "xrl_ppo_act_tail": [C.POINTER(PpoActTail), c_void_p],
    "xrl_linear_bwd_data": [C.POINTER(Gemm), c_int, c_void_p],
    "xrl_linear_bwd_weight": [C.POINTER(Gemm), c_int, c_int, c_int64, c_void_p],
    "xrl_ppo_loss_categorical": [C.POINTER(PpoLoss), c_void_p],
    "xrl_ppo_loss_gaussian": [C.POINTER(PpoLoss), c_void_p],
    "xrl_sum_partials": [c_void_p, c_int, c_int, c_void_p, c_void_p],
    "xrl_sum_partials_batched": [c_void_p, c_int, c_int, c_void_p, c_int, C.c_long, C.c_long, c_void_p],
    "xrl_grad_reduce": [c_void_p, c_int, c_int64, c_int64, c_void_p, c_void_p, c_int, c_void_p],
    "xrl_grad_reduce_fold": [c_void_p, c_int, c_int64, c_int64, c_void_p, c_void_p, c_int, c_int64, c_int, c_void_p],
    "xrl_adam_step": [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int, c_double, c_void_p],
    "xrl_adam_step_mirrored": [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int, c_double,
                               c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    "xrl_adam_step_mirrors": [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int, c_double,
                              C.POINTER(Mirrors), c_void_p],
    "xrl_reduce_adam": [c_void_p, c_int, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int,
                        c_double, C.POINTER(Mirrors), c_void_p, c_void_p],
    "xrl_reduce_adam_exchange": [c_void_p, c_int, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p,
                                 c_int, c_double, C.POINTER(Mirrors), c_void_p, C.POINTER(Exchange), c_void_p],
    "xrl_ipc_alloc": [C.c_size_t, C.POINTER(c_void_p), c_void_p],
    "xrl_ipc_open": [c_void_p, C.POINTER(c_void_p)],
    "xrl_ipc_close": [c_void_p],
    "xrl_ipc_free": [c_void_p],
    "xrl_ipc_clear": [c_void_p, C.c_size_t, c_void_p],
    "xrl_graph_begin": [c_void_p],
    "xrl_graph_end": [c_void_p, C.POINTER(c_void_p)],
    "xrl_graph_launch": [c_void_p, c_void_p],
    "xrl_graph_destroy": [c_void_p],
}

_lib = None


def exported_symbols():
    """Names every entry point include/xrl_hip.h declares (checked by the CPU test-suite)."""
    return ["xrl_version", "xrl_last_error", "xrl_rollout_cache_floats", "xrl_rollout_wide_words"] + list(_SIGS)


def load():
    """dlopen the in-tree library; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm must load ITS libamdhip64 first: libxrl_hip.so then binds to the same HIP runtime instance
    # (device memory, streams and kernels have to live in one runtime).
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise XrlError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(hipcc --offload-arch=gfx950). xuance_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    lib.xrl_version.restype = C.c_char_p
    lib.xrl_last_error.restype = C.c_char_p
    lib.xrl_rollout_cache_floats.restype = c_int64
    lib.xrl_rollout_cache_floats.argtypes = [C.POINTER(RolloutStep)]
    lib.xrl_rollout_wide_words.restype = c_int
    lib.xrl_rollout_wide_words.argtypes = []
    for name, args in _SIGS.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = c_int
    _lib = lib
    return lib


def check(rc, name=""):
    if rc != 0:
        raise XrlError(f"{name} failed (rc={rc}): {load().xrl_last_error().decode()}")


def call(name, *args):
    check(getattr(load(), name)(*args), name)


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    return None if t is None else t.data_ptr()


def stream_ptr():
    import torch
    return torch.cuda.current_stream().cuda_stream
