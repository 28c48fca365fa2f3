"""Drop-in for the reference ``options.py`` (MonodepthOptions, options.py:9-480): every flag name, type,
default and choice list is reproduced, including the reference's quirks — ``store_false`` flags whose default
is therefore True (``--need_4beam``, ``--need_2_channel``, ``--beam_encoder``, ``--trainer_siloss_all_scale``,
``--gdc_loss_only_on_scale_0``, ``--completion_siloss``) and "true"/"false" string switches.

Declared as a table rather than ~470 lines of ``add_argument`` calls; ``tests/test_options.py`` checks the
table against the reference's parser (defaults, types, choices) in the build container.
"""
import argparse

T, F = "store_true", "store_false"
TF = ["true", "false"]

# (flag, kind, default, choices)   kind: python type, or T / F for boolean switches; "+int"/"+str" = nargs="+"
FLAGS = [
    # paths
    ("data_path", str, "kitti_data", None), ("log_dir", str, "log", None),
    # training
    ("model_name", str, "mdp", None),
    ("split", str, "eigen_zhou", ["eigen_zhou", "eigen_full", "odom", "benchmark"]),
    ("num_layers", int, 50, [18, 34, 50, 101, 152]),
    ("dataset", str, "kitti", ["kitti", "kitti_odom", "kitti_depth", "kitti_test"]),
    ("png", T, None, None), ("height", int, 192, None), ("width", int, 640, None),
    ("disparity_smoothness", float, 1e-3, None), ("scales", "+int", [0, 1, 2, 3], None),
    ("min_depth", float, 0.1, None), ("max_depth", float, 100.0, None), ("use_stereo", T, None, None),
    ("frame_ids", "+int", [0, -1, 1], None),
    # optimisation
    ("batch_size", int, 5, None), ("learning_rate", float, 1e-4, None), ("num_epochs", int, 20, None),
    ("scheduler_step_size", int, 10, None),
    # ablation
    ("v1_multiscale", T, None, None), ("avg_reprojection", T, None, None), ("disable_automasking", T, None, None),
    ("predictive_mask", T, None, None), ("no_ssim", T, None, None),
    ("weights_init", str, "pretrained", ["pretrained", "scratch"]),
    ("pose_model_input", str, "pairs", ["pairs", "all"]),
    ("pose_model_type", str, "separate_resnet", ["posecnn", "separate_resnet", "shared"]),
    # system
    ("no_cuda", T, None, None), ("num_workers", int, 4, None),
    # loading
    ("load_weights_folder", str, "log/1337/models/weights_best/", None), ("train_load_weights_folder", str, None, None),
    ("refine_load_weights_folder", str, "log/mdp/models/weights_absrel7817/", None),
    ("models_to_load", "+str", ["encoder", "depth", "pose_encoder", "pose"], None),
    # logging
    ("log_frequency", int, 250, None), ("save_frequency", int, 1, None),
    # evaluation
    ("eval_stereo", T, None, None), ("eval_mono", T, None, None), ("disable_median_scaling", T, None, None),
    ("pred_depth_scale_factor", float, 1, None), ("ext_disp_to_eval", str, None, None),
    ("eval_split", str, "eigen", ["eigen", "eigen_benchmark", "benchmark", "odom_9", "odom_10"]),
    ("save_pred_disps", T, None, None), ("no_eval", T, None, None), ("eval_eigen_to_benchmark", T, None, None),
    ("eval_out_dir", str, None, None), ("post_process", T, None, None), ("eval_gdc", T, None, None),
    ("eval_batch_size", int, 1, None),
    # sparse-LiDAR fusion
    ("need_4beam", F, None, None), ("need_full_res_4beam", T, None, None), ("need_path", T, None, None),
    ("cat_4beam_to_color", T, None, None), ("need_2_channel", F, None, None), ("cat2start", T, None, None),
    ("cat2end", T, None, None), ("beam_encoder", F, None, None), ("trainer_siloss", str, "true", TF),
    ("trainer_siloss_all_scale", F, None, None), ("random_sample", int, -1, None),
    # refine
    ("train_entire_net", T, None, None), ("refine_shallow", T, None, None), ("refineUnet", T, None, None),
    ("refine_deep", T, None, None), ("refine_2d", T, None, None), ("refine_iter", int, 1, None),
    ("refine_iter_gama", float, 0.8, None), ("refine_offset", T, None, None),
    ("refine_depthnet_with_beam", str, "false", TF), ("clone_gdc", T, None, None), ("clone_path", str, None, None),
    ("need_inf_gdc", T, None, None), ("catxy", str, "true", TF), ("refine2d_deep", str, "true", TF),
    ("refine_a0", str, "true", TF), ("gdc_loss_threshold", float, 2.0, None), ("gdc_loss_weight", float, 0.008, None),
    ("gdc_loss_only_on_scale_0", F, None, None), ("gdc_abs_loss", float, 0.0, None), ("si_var", float, 0.3, None),
    # completion
    ("completion_val_split", str, "select", ["select", "full"]), ("completion_siloss_weight", float, 0.1, None),
    ("completion_siloss_all_scale", str, "false", TF), ("completion_eigen_crop", T, None, None),
    ("completion_num_epochs", int, 3, None), ("completion_scheduler_step_size", int, 25, None),
    ("completion_not_full_res", T, None, None), ("completion_amp", T, None, None),
    ("completion_pose_num_layers", int, 18, None), ("completion_siloss", F, None, None),
    ("completion_l1loss", T, None, None), ("completion_clip", float, 0.01, None),
    ("completion_num_layers", int, 50, [18, 34, 50, 101, 152]), ("completion_need2channel", str, "false", TF),
    ("completion_test", T, None, None),
    # debug / visualisation
    ("debug", T, None, None), ("visualize", T, None, None), ("vis_name", str, "diff", None),
    ("save_sample", int, -1, None), ("inf", T, None, None), ("demo", T, None, None),
    # depth-guided conv (unused by the trainer, kept for CLI parity)
    ("use_dropout", str, "true", TF), ("drop_channel", str, "true", TF), ("dropout_rate", float, 0.5, None),
    ("dropout_position", str, "early", ["early", "late", "adaptive"]), ("base_model", int, 50, None),
    ("adaptive_diated", str, "true", TF), ("deformable", str, "false", TF), ("use_rcnn_pretrain", str, "false", TF),
    ("d4twocha", str, "false", TF),
    # detection / evaluation extras
    ("det_name", str, None, None), ("per_semantic", T, None, None), ("run_name", str, None, None),
    ("nbeams", int, 4, None),
]


class MonodepthOptions:
    def __init__(self):
        self.parser = argparse.ArgumentParser(description="Monodepthv2 options")
        for name, kind, default, choices in FLAGS:
            kw = {}
            if kind in (T, F):
                kw["action"] = kind
            elif isinstance(kind, str) and kind.startswith("+"):
                kw.update(nargs="+", type={"int": int, "str": str}[kind[1:]], default=default)
            else:
                kw.update(type=kind, default=default)
                if choices is not None:
                    kw["choices"] = choices
            self.parser.add_argument("--" + name, **kw)

    def parse(self, args=None):
        self.options = self.parser.parse_args(args)
        return self.options
