for i in 1 2 3; do timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r4m_gputests_run$i.txt 2>&1; grep -E "passed|failed" gpurun_out/r4m_gputests_run$i.txt | tail -1; done
for i in 1 2 3; do if grep -q "failed" gpurun_out/r4m_gputests_run$i.txt; then grep -n "^E " gpurun_out/r4m_gputests_run$i.txt | head -30; fi; done
