#!/bin/bash
# Sanitizers over the host C code that runs under the reference's slice decoder -- the shim (shim/rcn_hip.c), the device DPB, the frame
# layer, the stream driver, the recorder -- and over the harness itself, driven on DRY frames (no GPU: the HIP half is the same object
# code, its device calls are not reached).  Build container only (needs /root/reference and oracle/_ref).
#   tools/san_dry.sh thread        8 frame threads, 33 pictures, twice on warm threads      -> "sanitizer reports: 0"
#   tools/san_dry.sh address       + tiles, a GOP-32 hierarchy on 16 threads, 1080p, 4K; AddressSanitizer fills fresh memory, so the
#                                  harness' own check "the device pass recorded what the record-only pass recorded" also shows any byte
#                                  of a recorded command that does not come from the stream
#   tools/san_dry.sh undefined
set -e
SAN=${1:-thread}
R=$(cd "$(dirname "$0")/.." && pwd); REF=${REF:-/root/reference}/libovvc; T=${TMPDIR:-/tmp}/ovvc_san_$SAN
[ -f $REF/slicedec.c ] || { echo "no reference tree: nothing to do"; exit 0; }
make -s -C $R/openvvc_amd/csrc && make -s -C $R/oracle
rm -rf $T && mkdir -p $T && cd $T
C=$R/openvvc_amd/csrc
for f in ovvc_record ovvc_record_dbf ovvc_record_intra ovvc_lmcs ovvc_md5 ovvc_calllog ovvc_dpb ovvc_frame ovvc_stream; do
  gcc -O1 -g -fPIC -fsanitize=$SAN -fno-omit-frame-pointer -pthread -I$R/include -I$C -c $C/$f.c -o $f.o
done
HIPOBJS=$(ls $C/build/*.o | grep -v -E "/ovvc_(record|record_dbf|record_intra|lmcs|md5|calllog|dpb|frame|stream)\.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=$SAN -o libovvc_hip.so $HIPOBJS *.o -lpthread -ldl 2>&1 | grep -v "not currently supported" || true
gcc -O1 -g -w -fPIC -shared -fsanitize=$SAN -pthread -I$REF -I$R/include -I$R/shim -DBITDEPTH=10 -o librcn_hip.so $R/shim/rcn_hip.c -L. -lovvc_hip -Wl,--allow-shlib-undefined
gcc -O1 -g -w -fPIC -shared -fsanitize=$SAN -I$REF -I$R/include -I$R/shim -DBITDEPTH=10 -o libgenpipe.so $R/oracle/ref_harness/gen_pipe.c \
    -L$R/oracle/_ref -lovvcref -L. -lrcn_hip -lovvc_hip -Wl,-z,lazy -lm -lpthread
gcc -O1 -fsanitize=$SAN -o gen_pipe $R/oracle/ref_harness/gen_pipe_main.c -L. -lgenpipe -Wl,--allow-shlib-undefined -Wl,-z,lazy
export LD_LIBRARY_PATH=$T:$R/oracle/_ref TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0" ASAN_OPTIONS=detect_leaks=0:halt_on_error=0 UBSAN_OPTIONS=print_stacktrace=1
run() { ./gen_pipe $T "$@" > run.log 2>&1 || { echo "gen_pipe $* failed:"; tail -5 run.log; exit 1; }
        n=$(grep -c -E "WARNING: ThreadSanitizer|ERROR: AddressSanitizer|runtime error" run.log || true); total=$((total + n))
        echo "gen_pipe $*: $n reports"; [ "$n" = 0 ] || grep -E -A6 "WARNING: ThreadSanitizer|ERROR: AddressSanitizer|runtime error" run.log | head -30; }
total=0
run device threads 8 pics 33 size 832 480 reps 2
run device threads 8 pics 17 size 832 480 cont 4
run device threads 8 pics 33 size 832 480 reps 2 bands 1          # band-wise submission: the DPB's row progress, bands left to later hooks
run device threads 16 pics 33 size 832 480 gop 32 bands 2 seed 11
if [ $SAN != thread ]; then
  run device threads 4 pics 5 seed 5 size 264 392 tiles 2 2
  run device threads 16 pics 33 size 832 480 gop 32 reps 2 seed 11
  run device pics 9 size 1920 1080 seed 31
  run device threads 4 pics 5 size 3840 2160
  run shim pics 5
fi
echo "sanitizer reports: $total"
