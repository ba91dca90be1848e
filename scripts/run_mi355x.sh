#!/bin/bash
# The stages of the reference's run.sh (equilibrate, restart + collect, parse, structural histograms) on the MI355X
# modules of this repository.  Same flags; the cluster flags of the reference (-c -nw -nt -mt ...) are accepted and ignored.
#   scripts/run_mi355x.sh [supercell=5] [pressures=32] [temperatures=32] [cycles=1024]
# For several GPUs start the first two stages under  python -m torch.distributed.run --nproc-per-node N -m neuralmelting_amd.remcmc ...
set -euo pipefail
s=${1:-5}; pn=${2:-32}; tn=${3:-32}; sn=${4:-1024}
root=$(cd "$(dirname "$0")/.." && pwd)
export PYTHONPATH="$root${PYTHONPATH:+:$PYTHONPATH}"
mkdir -p ./output/remcmc_$s
cd ./output/remcmc_$s
# equilibration run (nothing recorded: cutoff = number of cycles), restart dump at the end
python -m neuralmelting_amd.remcmc -v -n remcmc_init_$s -ss $s -bm -pn $pn -tn $tn -sn $sn -sc $sn -rd $sn
# data collection run, started from that dump
python -m neuralmelting_amd.remcmc -v -r -rn remcmc_init_$s -rs $sn -n remcmc_run_$s -ss $s -bm -pn $pn -tn $tn -sn $sn -rd $sn
# text -> arrays
python -m neuralmelting_amd.parse -v -n remcmc_run_$s
# radial and cartesian pair histograms
python -m neuralmelting_amd.distr -v -n remcmc_run_$s -cb 11
