#!/bin/bash
# ThreadSanitizer over the host C code that runs under several frame threads -- the shim (shim/rcn_hip.c), the device DPB, the frame
# layer, the stream driver, the recorder -- driven by the reference's slice decoder with 8 frame threads on DRY frames (no GPU: the
# HIP half is the same object code, its device calls are not reached).  Build container only (needs /root/reference and oracle/_ref).
#   tools/tsan_dry.sh            -> "ThreadSanitizer reports: 0"
set -e
R=$(cd "$(dirname "$0")/.." && pwd); REF=${REF:-/root/reference}/libovvc; T=${TMPDIR:-/tmp}/ovvc_tsan
[ -f $REF/slicedec.c ] || { echo "no reference tree: nothing to do"; exit 0; }
make -s -C $R/openvvc_amd/csrc && make -s -C $R/oracle
rm -rf $T && mkdir -p $T && cd $T
C=$R/openvvc_amd/csrc
for f in ovvc_record ovvc_record_dbf ovvc_record_intra ovvc_lmcs ovvc_md5 ovvc_calllog ovvc_dpb ovvc_frame ovvc_stream; do
  gcc -O1 -g -fPIC -fsanitize=thread -pthread -I$R/include -I$C -c $C/$f.c -o $f.o
done
HIPOBJS=$(ls $C/build/*.o | grep -v -E "/ovvc_(record|record_dbf|record_intra|lmcs|md5|calllog|dpb|frame|stream)\.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=thread -o libovvc_hip.so $HIPOBJS *.o -lpthread -ldl
gcc -O1 -g -w -fPIC -shared -fsanitize=thread -pthread -I$REF -I$R/include -I$R/shim -DBITDEPTH=10 -o librcn_hip.so $R/shim/rcn_hip.c -L. -lovvc_hip -Wl,--allow-shlib-undefined
gcc -O1 -g -w -fPIC -shared -fsanitize=thread -I$REF -I$R/include -I$R/shim -DBITDEPTH=10 -o libgenpipe.so $R/oracle/ref_harness/gen_pipe.c \
    -L$R/oracle/_ref -lovvcref -L. -lrcn_hip -lovvc_hip -Wl,-z,lazy -lm -lpthread
gcc -O1 -fsanitize=thread -o gen_pipe $R/oracle/ref_harness/gen_pipe_main.c -L. -lgenpipe -Wl,--allow-shlib-undefined -Wl,-z,lazy
LD_LIBRARY_PATH=$T:$R/oracle/_ref TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0" ./gen_pipe $T device threads 8 pics 33 size 832 480 reps 2 > run.log 2>&1 || { tail -5 run.log; exit 1; }
echo "ThreadSanitizer reports: $(grep -c 'WARNING: ThreadSanitizer' run.log)"
grep -o '"frames_differing": [0-9]*, "samples_differing": [0-9]*, "collocated_motion_entries_differing": [0-9]*' run.log | tail -1
