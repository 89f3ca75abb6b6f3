# builds proto-clip_amd/libpclip_<tag>.so with extra -D flags for pclip_encoder.hip:  tools/build_variants.sh tag1 "-DX=1" tag2 "-DY=2 -DZ" ...
cd "$(dirname "$0")/../proto-clip_amd/csrc" || exit 1
make -j8 >/dev/null 2>&1
while [ $# -ge 2 ]; do
  TAG=$1; FLAGS=$2; shift 2
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include $FLAGS -c pclip_encoder.hip -o /tmp/enc_$TAG.o 2>/tmp/enc_$TAG.err &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libpclip_$TAG.so pclip_api.o pclip_proto.o pclip_classify.o /tmp/enc_$TAG.o pclip_adapter.o pclip_resnet.o pclip_train.o pclip_preprocess.o ) &
done
wait
ls -la ../libpclip_*.so
