cd /root/repo
python3 tools/stage_times.py 64 2>&1 | tail -4
for g in 128 64 32; do echo "== grid $g"; CONAN_MEGA_GRID=$g python3 tools/stage_times.py 64 2>&1 | sed -n '2p;4p'; done
echo "== no mega"; CONAN_DEC_MEGA=0 python3 tools/stage_times.py 64 2>&1 | sed -n '2p;4p'
echo "== B=1"; python3 tools/stage_times.py 1 2>&1 | tail -4; CONAN_DEC_MEGA=0 python3 tools/stage_times.py 1 2>&1 | tail -4
