"""Command-line flags -- every flag name, type and default of the reference's ``src/arguments.py``
(Arguments 15-51, TrainArguments 85-118, TestArguments 124-133) so ``scripts/train.sh`` /
``scripts/sample.sh`` lines run unchanged, plus three additions: ``--dataset SyntheticDataset``,
``--precision {fp32,bf16}`` and ``--synthetic_len``.  ``parse()`` creates the same
``exp_dir/{checkpoints,logs,images}`` tree and appends ``args.txt`` (arguments.py:53-78)."""
import argparse
import os
from datetime import datetime

from . import dataset as dataset_module
from . import models as models_module
from .utils import get_modules, module_to_dict


class Arguments:
    def __init__(self):
        p = self.parser = argparse.ArgumentParser("Arguments for the program")
        p.add_argument("--dataroot", help="root folder of the dataset")
        p.add_argument("--name", type=str, default=f'{datetime.now().strftime("%Y-%m-%d_%H-%M-%S")}',
                       help="name of the experiment. It decides where to store samples and model")
        p.add_argument("--gpu_ids", type=str, default="0", help="gpu ids: e.g. 0  0,1,2, 0,2. use -1 for CPU")
        p.add_argument("--exp_dir", type=str, default="../exps", help="custom directory for storing experiment results")
        # model parameters
        p.add_argument("--model", type=str, default="BaseModel", help="chooses which model to use.")
        p.add_argument("--input_dim", type=int, default=3)
        p.add_argument("--output_dim", type=int, default=3)
        p.add_argument("--dim", type=int, default=64, help="# of gen filters in the last conv layer")
        p.add_argument("--init_type", type=str, default="normal", help="network initialization.")
        p.add_argument("--init_gain", type=float, default=0.02)
        p.add_argument("--use_dropout", action="store_true")
        p.add_argument("--num_domains", type=int, default=2, help="number of domains in the dataset")
        p.add_argument("--mode", type=str, default="train", help="train, val, test, etc")
        p.add_argument("--concat", action="store_true", help="concatenate style features for translation")
        p.add_argument("--reparam", action="store_true", help="reparameterize generating style features")
        p.add_argument("--use_dis_content", action="store_true", help="weather to use content discriminator")
        p.add_argument("--latent_dim", type=int, default=8, help="size of latent dimention")
        p.add_argument("--up_type", type=str, default="transpose", choices=["transpose", "nearest", "pixelshuffle"])
        p.add_argument("--dec_norm", type=str, default="layer", choices=["batch", "instance", "layer"])
        p.add_argument("--enc_norm", type=str, default="instance", choices=["batch", "instance", "layer"])
        # dataset parameters
        p.add_argument("--dataset", type=str, default="PairedDataset", choices=get_modules(dataset_module),
                       help="chooses how datasets are loaded.")
        p.add_argument("--shuffle", action="store_true")
        p.add_argument("--num_workers", default=4, type=int, help="# threads for loading data")
        p.add_argument("--batch_size", type=int, default=4, help="input batch size")
        p.add_argument("--load_size", type=int, default=286, help="scale images to this size")
        p.add_argument("--crop_size", type=int, default=256, help="then crop to this size")
        p.add_argument("--no_flip", action="store_true")
        p.add_argument("--select_domains", default=None, type=str, nargs="+")
        # additional parameters
        p.add_argument("--resume", type=str, default=None, help="path to checkpoint to load")
        p.add_argument("--save_logs", action="store_true")
        # additions of this implementation
        p.add_argument("--precision", type=str, default="fp32", choices=["fp32", "bf16"],
                       help="activation storage / MFMA input type (accumulation and statistics are always fp32)")
        p.add_argument("--seed", type=int, default=None,
                       help="(new, optional) base random seed; rank r of a multi-GPU run draws from seed + r")
        p.add_argument("--ckpt_module_prefix", action="store_true",
                       help="(new, optional) save model_*.ckpt with 'module.'-prefixed keys, as the reference's "
                            "DataParallel-wrapped GPU runs write and strictly expect them")
        p.add_argument("--hip_graph", action="store_true",
                       help="(new, optional) capture the training step into a hipGraph after 3 eager iterations and "
                            "replay it (single process only): removes the per-launch host cost, e.g. at --batch_size 1")
        p.add_argument("--synthetic_len", type=int, default=64, help="items per epoch of SyntheticDataset")

    def _finish(self, args):
        args.gpu_ids = [int(g) for g in args.gpu_ids.split(",") if int(g) >= 0]
        return args

    def parse(self, argv=None):
        args = self.parser.parse_args(argv)
        args.dataset = module_to_dict(dataset_module)[args.dataset]
        args.model = module_to_dict(models_module)[args.model]
        args.exp_dir = os.path.join(args.exp_dir, args.name)
        args.checkpoint_dir = os.path.join(args.exp_dir, "checkpoints")
        args.logdir = os.path.join(args.exp_dir, "logs")
        args.display_dir = os.path.join(args.exp_dir, "images")
        for d in (args.exp_dir, args.checkpoint_dir, args.logdir, args.display_dir):
            os.makedirs(d, exist_ok=True)
        self._finish(args)
        with open(os.path.join(args.exp_dir, "args.txt"), "a") as f:
            print("\n--- Loaded arguments ---")
            for name, value in sorted(vars(args).items()):
                print("%s: %s" % (str(name), str(value)))
                f.write("%s: %s\n" % (str(name), str(value)))
        return args


class TrainArguments(Arguments):
    def __init__(self):
        super().__init__()
        p = self.parser
        p.add_argument("--dis_norm", type=str, default=None, choices=["batch", "instance", "layer"])
        p.add_argument("--norm_feat", action="store_true")
        p.add_argument("--lr", type=float, default=0.0001)
        p.add_argument("--wd", type=float, default=0.0001)
        p.add_argument("--beta1", type=float, default=0.5)
        p.add_argument("--beta2", type=str, default=0.999)        # declared str in the reference (arguments.py:91)
        p.add_argument("--lr_policy", type=str, default="step")
        p.add_argument("--n_iters", type=int, default=1000000)
        p.add_argument("--last_iter", type=int, default=-1)
        p.add_argument("--max_iter", type=int, default=1000000)
        p.add_argument("--n_iter_decay", type=int, default=600000)
        p.add_argument("--d_iter", type=int, default=3)
        p.add_argument("--lambda_rec", type=float, default=10)
        p.add_argument("--lambda_cls", type=float, default=1.0)
        p.add_argument("--lambda_cls_G", type=float, default=5.0)
        p.add_argument("--lambda_style", type=float, default=5.0)
        p.add_argument("--print_freq", type=int, default=1000)
        p.add_argument("--save_freq", type=int, default=1000)
        p.add_argument("--display_freq", type=int, default=1000)
        p.add_argument("--train_n_batch", type=float, default=float("inf"))
        p.add_argument("--gan_mode", type=str, default="vanilla")
        p.add_argument("--resume_opt", type=str, default=None)
        p.add_argument("--ms_dis", action="store_true")
        p.add_argument("--dis_sn", action="store_true")
        p.add_argument("--num_scales", type=int, default=3)
        p.add_argument("--ms_dim", type=int, default=64,
                       help="(new, optional) width of the multi-scale discriminators; the reference hard-wires 64")
        p.add_argument("--use_ragan", action="store_true")
        p.add_argument("--lambda_perceptual", type=float, default=1.0)
        p.add_argument("--vgg_type", type=str, default="vgg19")
        p.add_argument("--vgg_loss", type=str, default=None)
        p.add_argument("--vgg_layers", type=str, nargs="+", default=["conv5_4"])
        p.add_argument("--layer_weights", type=float, nargs="+", default=[1.0])


class TestArguments(Arguments):
    def __init__(self):
        super().__init__()
        p = self.parser
        p.add_argument("--num", type=int, default=5, help="number of outputs per image")
        p.add_argument("--result_dir", type=str, default="./outputs")
        p.add_argument("--out_fmt", type=str, default="image")
        p.add_argument("--vid_fname", type=str, default="video.avi")
        p.add_argument("--reference", type=str, nargs="+", default=None)
        p.add_argument("--targets", type=str, nargs="+", default=None)
        p.add_argument("--multi_iter", type=int, default=0)
        p.add_argument("--save_visuals", action="store_true")
        p.add_argument("--gen_grid", action="store_true")
        p.add_argument("--gen_style", action="store_true")

    def parse(self, argv=None):
        args = self.parser.parse_args(argv)
        os.makedirs(args.result_dir, exist_ok=True)
        args.display_dir = os.path.join(args.result_dir, "videos" if "video" in args.out_fmt else "images")
        os.makedirs(args.display_dir, exist_ok=True)
        self._finish(args)
        args.mode = "test"
        args.dis_scale, args.dis_norm, args.dis_sn = 3, None, False
        args.ms_dis, args.use_ragan, args.vgg_loss, args.resume_opt = False, False, None, None
        args.model = module_to_dict(models_module)[args.model]
        with open(os.path.join(args.result_dir, "args.txt"), "a") as f:
            print("\n--- Loaded arguments ---")
            for name, value in sorted(vars(args).items()):
                print("%s: %s" % (str(name), str(value)))
                f.write("%s: %s\n" % (str(name), str(value)))
        return args
