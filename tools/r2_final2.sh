for wl in c1 c3 c4; do bash tools/pmc.sh r2d_$wl --workload $wl > /dev/null 2>&1; rm -rf gpurun_out/pmc_r2d_$wl; done
ls gpurun_out | grep summary
