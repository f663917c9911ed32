"""f1: the tensor names leopard_amd reads from a checkpoint are exactly the names the reference's converter
writes (toolkits/model_checkpoints_convertor/llava/hf2megatron_llava.py: the megatron -> HF direction, name map :168-175,
LLM layers :1150-1300, vision / projector / embeddings :1309-1455).  The converter's source is parsed for the string
fragments it assembles keys from — nothing is executed (it imports megatron) and nothing is copied.  Runs only where
/root/reference exists (the build container)."""
import os
import re

import pytest

CONV = "/root/reference/Pai-Megatron-Patch/toolkits/model_checkpoints_convertor/llava/hf2megatron_llava.py"
pytestmark = pytest.mark.skipif(not os.path.exists(CONV), reason="reference tree not present (GPU box)")


def test_engine_reads_exactly_the_converter_key_set():
    from leopard_amd.config import full_config
    from leopard_amd.synth import param_specs
    import io
    import tokenize
    src = open(CONV).read()
    lit = set()
    for tok in tokenize.generate_tokens(io.StringIO(src).readline):          # every string literal of the file, f-strings as written
        if tok.type == tokenize.STRING:
            body = re.sub(r"^[rbfuRBFU]*", "", tok.string)
            q = body[:3] if body[:3] in ('"""', "'''") else body[:1]
            lit.add(body[len(q):-len(q)])
    # fragments of the name map (:168-175) and of the vision branch (:1309-1416)
    llm_ops = {".self_attn.o_proj.", ".mlp.gate_proj.", ".mlp.up_proj.", ".mlp.down_proj.", ".input_layernorm.", ".post_attention_layernorm."}
    assert llm_ops <= lit
    assert "language_model.model.layers.{layer_idx}" in lit and ".self_attn.{QKV[index]}.weight" in lit
    assert re.search(r"QKV = \{0: 'q_proj', 1: 'k_proj', 2: 'v_proj'\}", src)
    vit_prefix = "vision_tower.vision_model.encoder.layers."
    vit_suffix = {".self_attn.out_proj.weight", ".self_attn.out_proj.bias", ".mlp.fc1.weight", ".mlp.fc1.bias", ".mlp.fc2.weight",
                  ".mlp.fc2.bias", ".layer_norm1.weight", ".layer_norm1.bias", ".layer_norm2.weight", ".layer_norm2.bias"}
    assert vit_prefix in lit and vit_suffix <= lit and ".self_attn.{QKV[index]}.{weight_or_bias}" in lit

    def written_by_converter(name: str) -> bool:
        if name in lit:                                               # embeddings, final norm, lm_head, projector, patch / position embedding, post LN
            return True
        m = re.fullmatch(r"language_model\.model\.layers\.(\d+)(\..+?\.)weight", name)
        if m:
            return m.group(2) in llm_ops or m.group(2) in {".self_attn.q_proj.", ".self_attn.k_proj.", ".self_attn.v_proj."}
        m = re.fullmatch(re.escape(vit_prefix) + r"(\d+)(\..+)", name)
        if m:
            return m.group(2) in vit_suffix or bool(re.fullmatch(r"\.self_attn\.[qkv]_proj\.(weight|bias)", m.group(2)))
        return False

    names = [n for n, _, _ in param_specs(full_config())]
    unknown = [n for n in names if not written_by_converter(n)]
    assert not unknown, unknown[:10]
    # and the other way round: every non-layer key the converter writes is consumed (or deliberately ignored, listed here)
    ignored = {"vision_tower.vision_model.pre_layrnorm.weight", "vision_tower.vision_model.pre_layrnorm.bias",      # CLIP tower only
               "vision_tower.vision_model.embeddings.cls_token", "vision_tower.vision_model.embeddings.class_embedding"}
    top = {l for l in lit if re.match(r"(language_model|multi_modal_projector|vision_tower\.vision_model)\.[a-z_.0-9]+\.(weight|bias)$", l)
           or l in ignored}
    assert top - ignored <= set(names), sorted(top - ignored - set(names))
    assert len(names) == 27 * 16 + 3 + 2 + 4 + 32 * 9 + 3            # SigLIP layers, patch/pos embeddings, post-LN, projector, Llama layers, embed/norm/head
