"""Training driver -- same loop as the reference's ``src/train.py:30-59``: per iteration
``update_lr -> set_inputs -> optimize_parameters(global_iter)``, then periodic loss logging, checkpoint
and image dump (all three fire at iteration 0, like the reference); stops after
``min(n_iters, max_iter)`` iterations with a final save.

    python -m masterthesis_amd.train --model AdaINModel --dataset SyntheticDataset --batch_size 8 ...
    torchrun --nproc-per-node 8 -m masterthesis_amd.train ...      (one process per GPU, RCCL)
"""
import torch

from .arguments import TrainArguments
from .distributed import init_from_env, rank_and_world
from .utils import TimerBlock


class Trainer:
    def load_dataset(self, args):
        with TimerBlock("Loading Dataset and creating dataloaders") as block:
            block.log("Create dataset object")
            dataset = args.dataset(args)
            block.log("Create dataloader")
            # the reference hard-codes shuffle=False (train.py:19); drop_last avoids the short last batch
            # that breaks torch.split(..., batch_size) there
            # one process per GPU: every rank iterates its own strided share of the indices (the reference's single
            # process feeds all GPUs from one loader through nn.DataParallel's scatter)
            _, world = rank_and_world()
            sampler = (torch.utils.data.distributed.DistributedSampler(dataset, shuffle=False, drop_last=True)
                       if world > 1 else None)
            dataloader = torch.utils.data.DataLoader(dataset, batch_size=args.batch_size, shuffle=False, sampler=sampler,
                                                     num_workers=args.num_workers, drop_last=True,
                                                     pin_memory=torch.cuda.is_available())
        return dataloader

    def create_model(self, args):
        with TimerBlock("Creating model") as block:
            model = args.model(args)
            block.log("Initialize model")
            model.initialize()
        return model

    def train(self, args, model, dataloader, rank=0):
        with TimerBlock("Training model") as block:
            global_iter = args.last_iter + 1 if args.resume_opt is not None else 0
            iterations = min(args.n_iters, args.max_iter)
            block.log(f"Running for {iterations} iterations")
            while True:
                for batch in dataloader:
                    model.update_lr()
                    model.set_inputs(batch)
                    model.optimize_parameters(global_iter)
                    if rank == 0:
                        if global_iter % args.print_freq == 0:
                            block.log("\n")
                            block.log(f"Iteration: {global_iter}, LR : {model.get_current_lr()}")
                            model.write_loss(global_iter)
                            block.log(model.print_losses())
                        if global_iter % args.save_freq == 0:
                            block.log(f"Saving model inside : {args.checkpoint_dir}")
                            model.save(global_iter)
                        if global_iter % args.display_freq == 0 and global_iter % args.d_iter == 0:
                            block.log("Writing images")
                            model.save_images(global_iter)
                    global_iter += 1
                    if global_iter > iterations:
                        if rank == 0:
                            block.log(f"Saving model inside : {args.checkpoint_dir}")
                            model.save(global_iter)
                        block.log("Finished training")
                        return

    def run(self, args):
        rank, world, local = init_from_env()
        if getattr(args, "seed", None) is not None:
            torch.manual_seed(args.seed)        # (the reference never seeds; ranks get seed + rank after initialize())
        dataloader = self.load_dataset(args)
        model = self.create_model(args)
        self.train(args, model, dataloader, rank)


def main(argv=None):
    args = TrainArguments().parse(argv)
    Trainer().run(args)


if __name__ == "__main__":
    main()
