mkdir -p gpurun_out/kdexp13
export LANDING_LIB=$PWD/landing-controller_amd/_var/lib_kddev3.so
for t in 0 1 2 3; do LANDING_KD_TRACE=$t timeout 300 python tools/dev/kd_slowest.py 812 8 datagen > gpurun_out/kdexp13/tr_$t.txt 2>&1; done
