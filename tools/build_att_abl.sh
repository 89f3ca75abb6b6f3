# builds proto-clip_amd/libpclip_att<N>.so for N in "$@" (compile-time ablations of the persistent attention kernel)
cd "$(dirname "$0")/../proto-clip_amd/csrc" || exit 1
make -j8 >/dev/null 2>&1
for N in "$@"; do
  # N = <abl>[s<stagger>], e.g. 1, 0s4, 1s6
  A=${N%%s*}; S=0; case $N in *s*) S=${N##*s};; esac
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -DPCLIP_ATT_ABL=$A -DPCLIP_ATT_STAGGER=$S -c pclip_encoder.hip -o /tmp/enc_att$N.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libpclip_att$N.so pclip_api.o pclip_proto.o pclip_classify.o /tmp/enc_att$N.o pclip_adapter.o pclip_resnet.o pclip_train.o pclip_preprocess.o
done
ls -la ../libpclip_att*.so
