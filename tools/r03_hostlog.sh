export MNF_LIB_PATH=$PWD/active-perception-using-neural-radiance-fields_amd/libmi355nerf_diag.so MNF_HOST_LOG=1
python tools/exp_split.py 2>&1 | grep -E "mnf jobs|exp_split" | grep -A1 -B0 "jobs" | tail -60
