"""Model / workload dimensions of the Leopard-LLaVA multi-image prefill path.

Every number here is taken from the reference (citations use the SURVEY.md abbreviations):

* tile size 364, patch 14, 27 SigLIP layers, FFN 4304:
  evaluations/models/llava_multiimg_siglip_anyres.py:26,61,378,394,
  Pai-Megatron-Patch/megatron_patch/model/llava/clip_encoder.py:318-351
* 2x2 pixel shuffle -> 169 tokens / tile, projector 4608 -> 4096 -> 4096 (gelu-erf):
  llava_multiimg_siglip_anyres.py:165-192
* LLM = Llama-3.1-8B: 32 layers, d 4096, 32 q / 8 kv heads, FFN 14336, rope theta 5e5 + llama3 scaling:
  Pai-Megatron-Patch/examples/llava/train_multiimg_llava_siglip.sh:73,86-93,
  toolkits/model_checkpoints_convertor/llava/hf2megatron_llava.py:1026-1052 (image_token_index 128200)
* tile budget 50, eos ids 128001/128009, max_new_tokens 128: llava_multiimg_siglip_anyres.py:387,448-452
"""
from __future__ import annotations

import dataclasses
import json
from dataclasses import dataclass, field
from typing import Optional


@dataclass
class VisionConfig:
    hidden_size: int = 1152
    intermediate_size: int = 4304
    num_hidden_layers: int = 27
    num_attention_heads: int = 16
    image_size: int = 364
    patch_size: int = 14
    num_channels: int = 3
    layer_norm_eps: float = 1e-6
    hidden_act: str = "gelu_pytorch_tanh"

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads

    @property
    def grid(self) -> int:
        return self.image_size // self.patch_size

    @property
    def num_patches(self) -> int:
        return self.grid * self.grid

    @property
    def patch_dim(self) -> int:
        return self.num_channels * self.patch_size * self.patch_size


@dataclass
class RopeScaling:
    """llama3 frequency scaling (reference: Megatron-LM-240603/megatron/core/models/common/embeddings/
    rotary_pos_embedding.py:48-83)."""
    factor: float = 8.0
    low_freq_factor: float = 1.0
    high_freq_factor: float = 4.0
    original_max_position_embeddings: int = 8192


@dataclass
class TextConfig:
    hidden_size: int = 4096
    intermediate_size: int = 14336
    num_hidden_layers: int = 32
    num_attention_heads: int = 32
    num_key_value_heads: int = 8
    vocab_size: int = 128256
    rms_norm_eps: float = 1e-5
    rope_theta: float = 500000.0
    rope_scaling: Optional[RopeScaling] = field(default_factory=RopeScaling)
    head_dim_override: Optional[int] = None
    pad_token_id: Optional[int] = None
    sliding_window: Optional[int] = None          # Mistral (Leopard-Idefics2): 4096

    @property
    def head_dim(self) -> int:
        return self.head_dim_override or self.hidden_size // self.num_attention_heads


@dataclass
class LeopardConfig:
    vision_config: VisionConfig = field(default_factory=VisionConfig)
    text_config: TextConfig = field(default_factory=TextConfig)
    image_token_index: int = 128200
    projector_hidden_act: str = "gelu"
    pixel_shuffle_factor: int = 2
    # attributes the reference forward() reads from self.config (llava_multiimg_siglip_anyres.py:248-260)
    output_attentions: bool = False
    output_hidden_states: bool = False
    use_return_dict: bool = True
    vision_feature_layer: int = -1
    vision_feature_select_strategy: str = "full"

    @property
    def tokens_per_tile(self) -> int:
        return self.vision_config.num_patches // (self.pixel_shuffle_factor ** 2)

    @property
    def projector_in(self) -> int:
        return self.vision_config.hidden_size * self.pixel_shuffle_factor ** 2

    # ---- (de)serialisation: HF-style config.json ("LlavaConfig" layout written by the reference's
    # converter, hf2megatron_llava.py:1050-1052) -------------------------------------------------
    def to_dict(self) -> dict:
        d = dataclasses.asdict(self)
        d.pop("use_return_dict")            # a read-only property of HF configs (serialised as "return_dict"): AutoConfig rejects the key
        d["return_dict"] = self.use_return_dict
        d["model_type"] = "llava"
        return d

    @classmethod
    def from_dict(cls, d: dict) -> "LeopardConfig":
        vc = dict(d.get("vision_config", {}))
        tc = dict(d.get("text_config", {}))
        vkeys = {f.name for f in dataclasses.fields(VisionConfig)}
        tkeys = {f.name for f in dataclasses.fields(TextConfig)}
        rs = tc.get("rope_scaling")
        if isinstance(rs, dict):
            rkeys = {f.name for f in dataclasses.fields(RopeScaling)}
            rs = RopeScaling(**{k: v for k, v in rs.items() if k in rkeys}) if rs.get(
                "rope_type", rs.get("type", "llama3")) == "llama3" else None
        tcfg = TextConfig(**{k: v for k, v in tc.items() if k in tkeys and k != "rope_scaling"})
        if "rope_scaling" in tc:
            tcfg.rope_scaling = rs
        top = {k: v for k, v in d.items()
               if k in {f.name for f in dataclasses.fields(cls)} and k not in ("vision_config", "text_config")}
        if "return_dict" in d:
            top["use_return_dict"] = bool(d["return_dict"])
        return cls(vision_config=VisionConfig(**{k: v for k, v in vc.items() if k in vkeys}),
                   text_config=tcfg, **top)

    def save(self, path: str) -> None:
        with open(path, "w") as f:
            json.dump(self.to_dict(), f, indent=1)

    @classmethod
    def load(cls, path: str) -> "LeopardConfig":
        with open(path) as f:
            return cls.from_dict(json.load(f))


def full_config() -> LeopardConfig:
    """Leopard-LLaVA as released: SigLIP-SO400M/14@364 + Llama-3.1-8B."""
    return LeopardConfig()


def tiny_config(vocab: int = 512) -> LeopardConfig:
    """Reduced-width configuration used for golden fixtures (SURVEY.md 8c vi): same op graph, tiny dims.
    56x56 tiles of 14x14 patches -> 4x4=16 ViT tokens -> 4 visual tokens per tile."""
    return LeopardConfig(
        vision_config=VisionConfig(hidden_size=64, intermediate_size=128, num_hidden_layers=2,
                                   num_attention_heads=4, image_size=56, patch_size=14),
        text_config=TextConfig(hidden_size=64, intermediate_size=128, num_hidden_layers=2,
                               num_attention_heads=4, num_key_value_heads=2, vocab_size=vocab,
                               rope_theta=500000.0, rope_scaling=RopeScaling(), pad_token_id=0),
        image_token_index=500,
    )


def mid_config() -> LeopardConfig:
    """Mid-size configuration that exercises the production kernel shapes (head dims 72 / 128, the
    ragged 676-token ViT sequences, the 4304 FFN width) at a depth that a CPU oracle finishes in
    seconds: full-width layers, 2 ViT + 2 LLM layers, 8k vocab."""
    cfg = LeopardConfig()
    cfg.vision_config.num_hidden_layers = 2
    cfg.text_config.num_hidden_layers = 2
    cfg.text_config.vocab_size = 8192
    cfg.image_token_index = 8000
    return cfg


# ==================================================================================================
# Leopard-Idefics2 (evaluations/models/idefics2_multiimg.py; Idefics2-8B base: NaViT SigLIP + perceiver + Mistral-7B)
# dimensions: Pai-Megatron-Patch/examples/idefics2/train_multiimg_idefics2.sh:80,93-102,190,234,239 and the public
# idefics2-8b config (perceiver 64 latents x 3 layers, 16 heads x 96, 4 kv heads; vision 70x70 position grid)
# ==================================================================================================
@dataclass
class PerceiverConfig:
    n_latents: int = 64
    depth: int = 3
    n_heads: int = 16
    head_dim: int = 96
    num_key_value_heads: int = 4
    rms_norm_eps: float = 1e-6


@dataclass
class Idefics2Config:
    vision_config: VisionConfig = field(default_factory=lambda: VisionConfig(image_size=980))
    text_config: TextConfig = field(default_factory=lambda: TextConfig(vocab_size=32003, rope_theta=10000.0,
                                                                      rope_scaling=None, sliding_window=4096))
    perceiver_config: PerceiverConfig = field(default_factory=PerceiverConfig)
    image_token_id: int = 32001
    longest_edge: int = 980                    # idefics2_multiimg.py:23-25 (do_image_splitting=False)

    # the LLM core of LeopardEngine reads these two names
    @property
    def image_token_index(self) -> int:
        return self.image_token_id

    @property
    def tokens_per_tile(self) -> int:
        return 1                               # the prompt already carries n_latents <image> ids per image

    def to_dict(self) -> dict:
        d = dataclasses.asdict(self)
        d["model_type"] = "idefics2"
        return d


def idefics2_full_config() -> Idefics2Config:
    return Idefics2Config()


def idefics2_tiny_config() -> Idefics2Config:
    """Golden-fixture configuration (same op graph, tiny dims): 4x4 position grid, 4 latents."""
    return Idefics2Config(
        vision_config=VisionConfig(hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4,
                                   image_size=56, patch_size=14),
        text_config=TextConfig(hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4,
                               num_key_value_heads=2, vocab_size=512, rope_theta=10000.0, rope_scaling=None,
                               sliding_window=4096, pad_token_id=0),
        perceiver_config=PerceiverConfig(n_latents=4, depth=2, n_heads=4, head_dim=16, num_key_value_heads=2),
        image_token_id=500, longest_edge=56)


def idefics2_mid_config() -> Idefics2Config:
    """Full-width layers (vision 1152 / 16x72, perceiver 16x96 over 4096, Mistral 32/8x128), reduced depth."""
    cfg = Idefics2Config()
    cfg.vision_config.num_hidden_layers = 2
    cfg.text_config.num_hidden_layers = 2
    cfg.text_config.vocab_size = 8192
    cfg.perceiver_config.depth = 2
    cfg.image_token_id = 8000
    return cfg
