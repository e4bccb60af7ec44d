bash tools/tail_profile.sh 2>&1 | tail -3
python tools/tail_profile.py c2 depth=1 2>&1 | tail -10
python tools/tail_profile.py c2 depth=0 2>&1 | tail -10
