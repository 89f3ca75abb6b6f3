"""Proto-CLIP-F-Q^T (reference main.qt.py): identical to main.py except that every training step takes its queries from a
shuffled, augmented image loader — `clip_model.encode_image(images)` under no_grad (main.qt.py:198-201) — instead of sampling
rows of the key bank, the alpha grid is left un-rounded (110-111) and checkpoints go under `best-alpha-beta/` (292, 327)."""
from . import main as _main
from .main import get_arguments, populate_cfg_using_args  # noqa: F401  (same CLI, main.qt.py:24-72)


def run_proto_clip(cfg, visual_memory_keys, visual_memory_values, val_features, val_labels, test_features, test_labels,
                   textual_memory_bank, clip_model, text_prompts, train_loader_F):
    return _main.run_proto_clip(cfg, visual_memory_keys, visual_memory_values, val_features, val_labels, test_features,
                                test_labels, textual_memory_bank, clip_model, text_prompts, train_loader_F=train_loader_F,
                                variant="qt")
