# builds proto-clip_amd/libpclip_<tag>.so with extra -D flags for one translation unit (SRC=pclip_linear | pclip_attention | pclip_layernorm | pclip_stem, default pclip_linear):  tools/build_variants.sh tag1 "-DX=1" tag2 "-DY=2 -DZ" ...
cd "$(dirname "$0")/../proto-clip_amd/csrc" || exit 1
make -j8 >/dev/null 2>&1
while [ $# -ge 2 ]; do
  TAG=$1; FLAGS=$2; shift 2
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include $FLAGS -c ${SRC:-pclip_linear}.hip -o /tmp/enc_$TAG.o 2>/tmp/enc_$TAG.err &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libpclip_$TAG.so /tmp/enc_$TAG.o $(ls *.o | grep -v stress | grep -v "^${SRC:-pclip_linear}.o") ) &
done
wait
ls -la ../libpclip_*.so
