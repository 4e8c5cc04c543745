python tools/build_time.py > gpurun_out/r04o_build_default.json 2> gpurun_out/r04o.err
for b in 14 15; do SEMADB_AMD_LIB=$PWD/build/dc$b/libsemadb_amd.so python tools/build_time.py > gpurun_out/r04o_build_dc$b.json 2>> gpurun_out/r04o.err; done
