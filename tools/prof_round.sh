set -u
bash tools/prof_stats.sh l1 > /dev/null
bash tools/prof_stats.sh l6 --level 6 --blocks 2048 > /dev/null
bash tools/prof_stats.sh l12w --level 12 --block 32768 --blocks 8192 --corpus weblog > /dev/null
bash tools/prof_stats.sh l12 --level 12 --blocks 2048 > /dev/null
bash tools/prof_stats.sh l3 --level 3 --blocks 4096 > /dev/null
bash tools/prof_stats.sh l1rep --level 0x101 --blocks 4096 > /dev/null
bash tools/prof_pmc.sh l1 > /dev/null
bash tools/prof_pmc.sh l6 --level 6 --blocks 2048 > /dev/null
bash tools/prof_pmc.sh l12w --level 12 --block 32768 --blocks 8192 --corpus weblog > /dev/null
for t in l1 l6 l12w l12 l3 l1rep; do echo "== stats $t"; find gpurun_out/stats_$t -name '*kernel_stats.csv' -exec cat {} \; | head -4; done
for t in l1 l6 l12w; do echo "== pmc $t"; cat gpurun_out/pmc_$t/summary.txt; done
