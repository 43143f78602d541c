cd /root/repo
for g in 16 32 64 128; do echo "== single-tile mega grid $g"; CONAN_MEGA_SINGLE=1 CONAN_MEGA_GRID=$g python3 tools/mega_probe.py 1 2>&1 | grep -E "alone|last launch"; done
echo "== separate"; python3 tools/mega_probe.py 1 2>&1 | grep -E "alone"
