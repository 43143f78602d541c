cd /tmp && export TMPDIR=/tmp
cd /root/repo
cp conan_amd/libconan_hip.so /tmp/new.so; cp conan_amd/libconan_hip_st.so conan_amd/libconan_hip.so
CONAN_RB_NOPAIR=1 CONAN_CL_SHAPE=$1 python3 tools/blocking_trace.py 64 2>&1 | grep "conv_limb" | tail -10
cp /tmp/new.so conan_amd/libconan_hip.so
