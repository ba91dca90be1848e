"""helper launched by test_multiproc.py under torch.distributed.run (gloo, CPU): the driver with the oracle stand-in"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)


def main():
    import torch.distributed as dist
    from helpers import OracleEngine
    from neuralmelting_amd import remcmc
    from oracle import oracle as O
    cwd = sys.argv[1]
    argv = sys.argv[2:]
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    dist.init_process_group('gloo')
    run = remcmc.Run(argv, cwd=cwd, rank=rank, world=world)
    run.make_engine = lambda: OracleEngine(O, run)
    remcmc.run_guarded(run)


if __name__ == '__main__':
    main()
