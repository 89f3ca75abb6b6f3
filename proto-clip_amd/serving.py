"""Serving-style entry: the consumer of trained banks.  Mirrors the arithmetic of the
reference toolkit's `load_pretrained_mb_and_adapters` (toolkit/proto_clip_toolkit/utils/model_utils.py:12-67)
and `ProtoClipClassifier._load_trained_models_and_embeddings` / `classify_objects`
(toolkit/proto_clip_toolkit/ros/utils/proto_clip_classifier.py:48-71, 132-147): prototypes are formed once
from the saved banks, each request is encode_image -> row normalise -> adapter -> row normalise -> P -> top-k.

The robot demo calls it with batches of 1-8 crops, where the 150-odd kernel launches of a ViT forward are
launch-bound; `capture(batch)` records the whole request into a hipGraph (torch.cuda.CUDAGraph drives the
capture; every launch goes through the C ABI on the capture stream, which never allocates or syncs) and
`classify` replays it for that batch size.  ROS / OpenCV glue of the toolkit is out of scope."""
import os

import torch

from . import ops
from .model import Adapter, Adapter_FC
from .utils import get_model_dir_root


def load_pretrained_mb_and_adapters(config=None, memory_bank_v_path=None, memory_bank_t_path=None, adapter_type=None,
                                    adapter_weights_path=None):
    """(embeddings_v [N*K, D], embeddings_t [N, D], adapter) from a run directory (`config`) or explicit paths.
    Unlike the reference (which dereferences config['adapter'] when config is None, model_utils.py:55), the
    explicit-path form works for conv adapters too."""
    with torch.no_grad():
        if config:
            model_dir = f"{get_model_dir_root(config)}/alpha-beta/{config['alpha']}-{config['beta']}"
            prefix = f"best_lr_{config['lr']}_aug_{config['augment_epoch']}_epochs_{config['train_epoch']}"
            memory_bank_v_path = os.path.join(model_dir, f"{prefix}_v.pt")
            memory_bank_t_path = os.path.join(model_dir, f"{prefix}_t.pt")
            adapter_weights_path = os.path.join(model_dir, f"{prefix}_a.pt")
            adapter_type = config["adapter"]
        try:
            embeddings_v = torch.load(memory_bank_v_path, map_location="cuda")
            embeddings_t = torch.load(memory_bank_t_path, map_location="cuda")
        except Exception:
            raise FileNotFoundError(f"File does not exist: {memory_bank_v_path} and {memory_bank_t_path}")
        if adapter_type is None:
            raise Exception("Please mention the adapter type in the args or in the config file.")
        ndim = embeddings_v.shape[1]
        adapter = (Adapter(ndim, c_type=adapter_type, dtype=torch.half) if "conv" in adapter_type
                   else Adapter_FC(ndim, dtype=torch.half)).cuda()
        try:
            adapter.load_state_dict(torch.load(adapter_weights_path, map_location="cuda"))
        except Exception:
            raise FileNotFoundError(f"File does not exist: {adapter_weights_path}")
    return embeddings_v.detach(), embeddings_t.detach(), adapter


class ProtoClipClassifier:
    """classify(images [B,3,R,R]) -> (top-k probabilities [B,k] fp32, top-k class indices [B,k] int64)."""

    def __init__(self, clip_model, embeddings_v, embeddings_t, adapter, shots, alpha, beta, top_k=5, class_names=None, low_latency=True,
                 auto_graph=False, max_graphs=8):
        self.clip_model, self.adapter = clip_model, adapter
        self.low_latency = bool(low_latency)       # split-K linears for small requests (ops.low_latency)
        self.auto_graph, self.max_graphs = bool(auto_graph), int(max_graphs)   # capture a hipGraph per new batch size on first use
        self.alpha, self.beta, self.top_k = float(alpha), float(beta), int(top_k)
        self.class_names = class_names
        NxK = embeddings_v.shape[0]
        self.N = NxK // shots
        with torch.no_grad():                              # proto_clip_classifier.py:58-71
            self.z_img_proto, self.zi_sq = ops.proto_build(embeddings_v.detach(), self.N, shots, want_sq=True)
            self.z_text_proto, self.zt_sq = ops.l2norm_rows(embeddings_t.detach(), want_sq=True)
        self._graphs = {}

    def _forward(self, images):
        with torch.no_grad(), ops.low_latency(self.low_latency):
            f = self.clip_model.encode_image(images)                       # model_utils.py:75-77
            f = ops.l2norm_rows(f, out=f)
            a, a_sq = self.adapter_forward(f)                              # proto_clip_classifier.py:141-142
            _, _, tp, ti = ops.classify(a, self.z_img_proto, self.z_text_proto, self.alpha, self.beta, want_p=False,
                                        want_argmax=False, topk=self.top_k, q_sq=a_sq, zi_sq=self.zi_sq, zt_sq=self.zt_sq)
        return tp, ti

    def adapter_forward(self, f):
        if isinstance(self.adapter, Adapter):
            ad = self.adapter
            return ops.adapter_conv(f, ad.c_type == "conv-3x", ad.conv1.weight, ad.bn1.weight, ad.bn1.bias, ad.conv2.weight,
                                    ad.bn2.weight, ad.bn2.bias, ad.conv3.weight, ad.bn3.weight, ad.bn3.bias, l2norm_out=True,
                                    want_sq=True)
        fc = self.adapter.fc
        return ops.adapter_fc(f, fc[0].weight, fc[1].weight, fc[1].bias, fc[2].weight, fc[3].weight, fc[3].bias, ratio=0.2,
                              l2norm_out=True, want_sq=True)

    def capture(self, batch_size: int, warmup: int = 2):
        """Record one request of `batch_size` images into a hipGraph; later calls of that size replay it."""
        res = self.clip_model.visual.input_resolution
        static_in = torch.zeros(batch_size, 3, res, res, dtype=torch.float32, device="cuda")
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                      # warm-up off the capture: attribute set-up, scratch growth
            for _ in range(warmup):
                self._forward(static_in)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        ops.evict_workspace(side)                          # the warm-up stream's scratch buffer is not needed again
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            tp, ti = self._forward(static_in)
        self._graphs[batch_size] = (g, static_in, tp, ti)
        return self

    def classify(self, images):
        B = images.shape[0]
        hit = self._graphs.get(B)
        if hit is None and self.auto_graph and images.dtype == torch.float32 and len(self._graphs) < self.max_graphs and B > 0:
            self.capture(B)                                # eager launches cannot keep up with a small request
            hit = self._graphs.get(B)
        if hit is not None and images.dtype == torch.float32:
            g, static_in, tp, ti = hit
            static_in.copy_(images, non_blocking=True)
            g.replay()
            return tp.clone(), ti.long()
        tp, ti = self._forward(images)
        return tp, ti.long()

    def classify_objects(self, images):
        """Reference surface (proto_clip_classifier.py:132-158): (top-k class names, top-k probabilities)."""
        tp, ti = self.classify(images)
        if self.class_names is None:
            return ti, tp
        names = [[self.class_names[int(x)].replace("_", " ") for x in row] for row in ti.cpu()]
        return names, tp


def test_ood_performance(cfg, test_loader, clip_model=None, memory_bank_v_path=None, memory_bank_t_path=None, adapter_type=None,
                         adapter_weights_path=None):
    """Accuracy (percent) of trained banks + adapter on an out-of-distribution test split (reference toolkit
    ood_utils.py:58-111).  The reference builds the ImageNet-V2 / ImageNet-Sketch datasets itself (torchvision /
    imagenetv2_pytorch: out of scope); here the caller passes the loader of pre-processed `[B,3,R,R]` batches."""
    from .utils import pre_load_features
    if clip_model is None:
        from . import clip
        clip_model, _ = clip.load(cfg["backbone"])
    clip_model.eval()
    os.makedirs(get_model_dir_root(cfg), exist_ok=True)                                               # feature cache directory
    test_features, test_labels = pre_load_features(cfg, "test", clip_model, test_loader)             # ood_utils.py:83
    with torch.no_grad():
        embeddings_v, embeddings_t, adapter = load_pretrained_mb_and_adapters(
            memory_bank_v_path=memory_bank_v_path, memory_bank_t_path=memory_bank_t_path, adapter_type=adapter_type,
            adapter_weights_path=adapter_weights_path)
        K = cfg["shots"]
        N = embeddings_v.shape[0] // K
        z_text_proto = ops.l2norm_rows(embeddings_t)                                                 # 101-102
        feats = adapter(test_features, l2norm_out=True)                                              # 104-105
        # 96-99 (prototypes) + 107-109 (P, argmax): one launch where the class count allows it, the two calls otherwise (same bits)
        _, _, am, _, _ = ops.proto_classify(embeddings_v, N, K, feats, z_text_proto, cfg["alpha"], cfg["beta"])
        correct = (am.long() == test_labels.to(am.device)).sum().item()
    return 100.0 * correct / max(test_features.shape[0], 1)
