"""helper launched by test_multiproc.py under torch.distributed.run (gloo, CPU): bench.py's main with the oracle stand-in as the engine,
so that the N > 1 legs (strong = the preset's own grid dealt out over the ranks, weak = N times as many rows) and their reductions
run without a GPU.  Prints bench.py's one JSON line on rank 0."""
import functools
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)


def main():
    os.environ['NM_BENCH_BACKEND'] = 'gloo'
    from helpers import BenchOracleEngine
    from oracle import oracle as O
    import bench
    O.build()
    bench.main(sys.argv[1:], make_engine=functools.partial(BenchOracleEngine, O))


if __name__ == '__main__':
    main()
