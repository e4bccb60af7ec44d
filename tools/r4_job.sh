S=r4_k_extra
timeout 1200 python tests/tools/fuzz_parity.py 600 5101 > gpurun_out/${S}_fuzz_parity_600_scenes.txt 2>&1; tail -1 gpurun_out/${S}_fuzz_parity_600_scenes.txt
timeout 1200 python tests/tools/fuzz_parity.py 400 5102 --queue > gpurun_out/${S}_fuzz_parity_400_scenes_queue.txt 2>&1; tail -1 gpurun_out/${S}_fuzz_parity_400_scenes_queue.txt
timeout 1200 python tests/tools/fuzz_parity.py 300 5103 --queue --spheres > gpurun_out/${S}_fuzz_parity_300_scenes_queue_spheres.txt 2>&1; tail -1 gpurun_out/${S}_fuzz_parity_300_scenes_queue_spheres.txt
timeout 1200 python tests/tools/fuzz_parity.py 200 5104 --queue --bare-spheres > gpurun_out/${S}_fuzz_parity_200_scenes_queue_bare_spheres.txt 2>&1; tail -1 gpurun_out/${S}_fuzz_parity_200_scenes_queue_bare_spheres.txt
timeout 1200 python tests/tools/fuzz_parity.py 250 5105 --kind=merl --merl-tiers --spheres > gpurun_out/${S}_fuzz_parity_250_scenes_measured_brdf_both_tiers.txt 2>&1; tail -1 gpurun_out/${S}_fuzz_parity_250_scenes_measured_brdf_both_tiers.txt
