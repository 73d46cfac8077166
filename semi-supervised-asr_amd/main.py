"""Same CLI as the reference main.py:6-41 (flags -config/-c, --sup_pretrain, --judge_pretrain, --ssl_train,
--load_model, --load_judge, --test); yaml.safe_load because bare yaml.load() raises on PyYAML 6.
Launch under torch.distributed.run for data parallelism (one process per GPU)."""
from argparse import ArgumentParser

import yaml

from solver import Solver

if __name__ == "__main__":
    parser = ArgumentParser()
    parser.add_argument("-config", "-c", default="config.yaml")
    for flag in ("--sup_pretrain", "--judge_pretrain", "--ssl_train", "--load_model", "--load_judge", "--test"):
        parser.add_argument(flag, action="store_true")
    args = parser.parse_args()
    with open(args.config, "r") as f:
        config = yaml.safe_load(f)
    solver = Solver(config, load_model=args.load_model)
    if args.load_judge:
        solver.load_judge(config["load_judge_path"], config["load_optimizer"])
    if args.sup_pretrain:
        solver.sup_pretrain()
    if args.judge_pretrain:
        solver.judge_pretrain()
    if args.ssl_train:
        solver.ssl_train()
    if args.test:
        solver.test()
