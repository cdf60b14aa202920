#!/bin/bash
# Register / scratch / occupancy of every kernel instantiation of the library (hipcc -Rpass-analysis=kernel-resource-usage over all
# translation units, in parallel).  Usage: bash tools/kernel_resources_all.sh > profiles/rNN_kernel_resources.txt
REPO=$(cd $(dirname $0)/.. && pwd)
cd $REPO/bp_osd_amd/csrc
T=$(mktemp -d)
for f in *.hip; do
  (hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -fvisibility=hidden -c $f -o $T/${f%.hip}.o -Rpass-analysis=kernel-resource-usage 2> $T/${f%.hip}.txt) &
done
wait
for f in $T/*.txt; do python $REPO/tools/kernel_resources.py $f | sed 's/ \[-Rpass-analysis=kernel-resource-usage\]//'; done
rm -rf $T
