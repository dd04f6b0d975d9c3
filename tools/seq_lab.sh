# Lab builds of the library with a SEQ_EXP timing switch in k_block_x6<.., SEQ> (results are then WRONG): tools/seqlab_<n>.so.
# usage: bash tools/seq_lab.sh 1 2 3 ...   (run here, in the build container; the .so files travel to the GPU box)
set -e
cd "$(dirname "$0")/.."
OBJ=influentialrs_amd/csrc/_obj
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -DIRS_LAB -DSEQ_EXP=$n -c influentialrs_amd/csrc/decoder.hip -o /tmp/decoder_seqlab_$n.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/seqlab_$n.so $OBJ/capi.o /tmp/decoder_seqlab_$n.o $OBJ/score.o $OBJ/path.o $OBJ/comm.o -ldl
  echo built tools/seqlab_$n.so
done
