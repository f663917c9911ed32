"""Evaluation-harness counterpart of the reference's ``run_llava_local_inference``
(evaluations/models/llava_multiimg_siglip_anyres.py:364-500, "EVAL"), as separable functions.

The reference harness is one inline loop; its Python never travels to the GPU box, so this module reproduces,
for the same input record ``{images_path, question, answers, ques_type, options}`` + setting:
  (i)   the exact prompt string                                  EVAL:408-442
  (ii)  the ViT-input order and count                            EVAL:386-405
  (iii) the generate() keyword arguments                         EVAL:448-452
  (iv)  the result-row schema and the shard file name            EVAL:480-484, 496-497
including the published script's quirks.  It is pinned against a capture of the reference harness
itself (tests/golden/harness_capture.json, produced by oracle/gen_golden.py with mocked loaders).

Scoring (eval_utils.py parse_*/eval_*) is CPU string processing outside the hot path: a ``scorer`` callable is
injected; when none is given the row carries the raw response only.
"""
from __future__ import annotations

import json
import os
import re
from dataclasses import dataclass
from typing import Callable, Dict, List, Optional, Sequence

from .tiler import SAMPLE_BUDGET, TILE, cut_tiles, choose_canvas, letterbox, plan_tile_budget

HEAD = "<|begin_of_text|><|start_header_id|>user<|end_header_id|>\n\n"        # EVAL:22
TAIL = "<|eot_id|><|start_header_id|>assistant<|end_header_id|>\n\n"           # EVAL:23
IMAGE_TAG = "<image>"
TOK_OPEN, TOK_IMG, TOK_CLOSE = "<|reserved_special_token_20|>", "<|reserved_special_token_195|>", "<|reserved_special_token_21|>"
EOS_IDS = [128001, 128009]                                                     # EVAL:450
MAX_NEW_TOKENS = 128                                                           # EVAL:452
MAX_PROMPT_TOKENS = 16384                                                      # EVAL:443


def get_instruction(setting: str, ques_type: str) -> str:
    """evaluations/models/eval_utils.py:104-119."""
    table = {
        True: {"cot": "First think step by step. Then answer with the letter of the correct option.",
               "direct": "Answer with the option's letter from the given choices directly.", "none": ""},
        False: {"cot": "First think step by step. Then answer with a single word or phrase.",
                "direct": "Answer the question using a single word or phrase.", "none": ""},
    }
    return table[ques_type == "multiple-choice"][setting]


def split_shard(rows: list, shard: int, num_shards: int) -> list:
    """evaluations/models/eval_utils.py:84-89 (contiguous slices of len//num_shards + 1)."""
    size = len(rows) // num_shards + 1
    return rows[shard * size:(shard + 1) * size]


def keep_first_image_tags(text: str, n: int) -> str:
    """eval_utils.retain_n_images (:122-146): keep the first n ``<image>`` tags, drop the rest."""
    seen = 0

    def repl(m):
        nonlocal seen
        seen += 1
        return m.group(0) if seen <= n else ""
    return re.sub(re.escape(IMAGE_TAG), repl, text)


@dataclass
class PreparedSample:
    prompt: str
    question: str                 # the (possibly tag-patched) question stored in the result row
    vit_inputs: list              # PIL images in ViT order: per image [original] + tiles
    tiles_per_image: List[int]    # what the prompt builder believes (EVAL "num_patchs_per_images_real")
    n_image_tokens: int


def tile_images(images: Sequence) -> (list, List[int]):
    """EVAL:386-401."""
    budget = SAMPLE_BUDGET - len(images)
    if budget <= 0:
        return list(images), [1] * len(images)              # sic (EVAL:400-401)
    allowance = plan_tile_budget([im.size for im in images], TILE, budget)
    vit_inputs, real = [], []
    for im, n in zip(images, allowance):
        canvas = choose_canvas(im.size, n, TILE)
        tiles = cut_tiles(letterbox(im, canvas), TILE) if canvas is not None else []
        real.append(len(tiles))
        vit_inputs.append(im)
        vit_inputs.extend(tiles)
    return vit_inputs, real


def build_prompt(question: str, n_images: int, tiles_per_image: List[int], setting: str, ques_type: str):
    """EVAL:408-442, quirks included: the tag count is taken BEFORE missing tags are prepended and is what
    drives the number of image-token groups; more tags than images indexes past the tile list (IndexError,
    exactly as the reference does)."""
    instruction = get_instruction(setting, ques_type)
    tag_count = question.count(IMAGE_TAG)
    if tag_count < n_images:
        question = f"{IMAGE_TAG * (n_images - tag_count)} {question}"
    elif tag_count > n_images:
        question = keep_first_image_tags(question, tag_count - n_images)         # sic (EVAL:418-419)
    prompt = f"{HEAD}{question}\n{instruction}{TAIL}"
    groups = []
    for k in range(tag_count):
        per_image = 1
        if tiles_per_image:
            per_image += tiles_per_image[k]                                      # IndexError if k >= len (reference too)
        groups.append(f"image {k + 1}: {TOK_OPEN}{TOK_IMG * per_image}{TOK_CLOSE}")
    for gtxt in groups:
        prompt = prompt.replace(IMAGE_TAG, gtxt, 1)
    prompt = prompt.replace("\r\n\t\t\r\n\t\t", " ")                             # EVAL:442
    return prompt, question, sum(g.count(TOK_IMG) for g in groups)


def prepare_sample(record: dict, setting: str, open_image: Optional[Callable] = None, pixels: bool = True) -> PreparedSample:
    """``pixels=False``: plan only — ``vit_inputs`` then holds the source images and the tile cutting is left to the GPU tiler
    (leopard_amd.gpu_tiler), which produces the very tiles ``tile_images`` would."""
    from PIL import Image
    opener = open_image or (lambda p: Image.open(p).convert("RGB"))
    images = [opener(p) for p in record["images_path"]]
    if pixels:
        vit_inputs, real = tile_images(images)
    else:
        from .tiler import plan_sample
        budget = SAMPLE_BUDGET - len(images)
        real = [1] * len(images) if budget <= 0 else plan_sample([im.size for im in images]).tiles_per_image     # sic (EVAL:400-401)
        vit_inputs = images
    prompt, question, n_tok = build_prompt(record["question"], len(images), real, setting, record["ques_type"])
    return PreparedSample(prompt, question, vit_inputs, real, n_tok)


def plan_record(record: dict, setting: str, tokenizer, open_image: Optional[Callable] = None):
    """What a record needs BEFORE its pixels: (question for the result row, number of ViT inputs, prompt token ids).  Image files are opened
    for their sizes only (PIL reads the header; nothing is decoded), the tile counts come from the integer plan (leopard_amd.tiler.plan_sample
    == what cutting the tiles would give, EVAL:386-401), the prompt from build_prompt (EVAL:408-442)."""
    from PIL import Image
    from .tiler import plan_sample
    sizes = []
    for p in record["images_path"]:
        im = open_image(p) if open_image else Image.open(p)
        sizes.append(im.size)
        if open_image is None:
            im.close()
    budget = SAMPLE_BUDGET - len(sizes)
    real = [1] * len(sizes) if budget <= 0 else plan_sample(sizes).tiles_per_image                                  # sic (EVAL:400-401)
    prompt, question, _ = build_prompt(record["question"], len(sizes), real, setting, record["ques_type"])
    n_vit = len(sizes) if budget <= 0 else len(sizes) + sum(real)
    enc = tokenizer([prompt], return_tensors="pt", truncation=True, max_length=MAX_PROMPT_TOKENS)["input_ids"]
    return question, n_vit, enc


def generate_kwargs(pad_token_id) -> dict:
    """EVAL:448-452."""
    return {"pad_token_id": pad_token_id, "eos_token_id": list(EOS_IDS), "max_new_tokens": MAX_NEW_TOKENS, "use_cache": True}


def result_row(record: dict, question: str, response: str, n_vit_inputs: int, scorer: Optional[Callable] = None) -> dict:
    """EVAL:456-484.  ``multi_img`` follows the reference: it tests the ViT-input list (thumbnails + tiles)."""
    correct, chosen, anls = None, response, 0
    if scorer is not None:
        correct, chosen, anls = scorer(record, response)
    return {"correct": correct, "chosen": chosen, "gold": record["answers"], "raw": response, "question": question,
            "image_type": record.get("image_type", None), "multi_img": n_vit_inputs > 1, "correct_anls": anls}


def shard_result_path(checkpoint: str, shard: int, setting: str, dataset: str) -> str:
    return os.path.join(checkpoint, f"{shard}_{setting}_{dataset}_shard_details.jsonl")      # EVAL:496-497


def run_inference(records: List[dict], model, tokenizer, setting: str = "direct", scorer: Optional[Callable] = None,
                  device=None, gpu_tiler=None, batch_size: int = 1, stats: Optional[dict] = None) -> List[dict]:
    """The hot loop (EVAL:381-487) over already-sharded records, greedy.  ``gpu_tiler`` (a leopard_amd.gpu_tiler.GpuTiler):
    resize / pad / crop run on the device and the u8 tile stack goes straight to the model (same pixels as the PIL path, bit
    for bit); without it the reference's host pipeline is used.  ``batch_size`` > 1: that many decode SLOTS are kept busy over all
    records (``model.generate_stream``: continuous batching — one captured decode step per token for all slots, a finished record's slot
    goes to the next record at once) instead of the reference's one ``generate`` per record; the rows are the same, in record order.
    ``stats`` (dict) receives the slot occupancy.  A model without ``generate_stream`` falls back to fixed groups (``generate_batch``)."""
    import numpy as np
    import torch
    from .tiler import siglip_preprocess
    rows = []
    size = getattr(getattr(getattr(model, "config", None), "vision_config", None), "image_size", TILE)    # 364 for Leopard
    dev = device if device is not None else model.device

    def prepare(rec):
        if gpu_tiler is not None:
            s = prepare_sample(rec, setting, pixels=False)
            pixel_values, plan = gpu_tiler.tile_sample([np.asarray(im.convert("RGB"), dtype=np.uint8) for im in s.vit_inputs])
            n_vit = plan.n_vit_inputs
        else:
            s = prepare_sample(rec, setting)
            pixel_values = torch.from_numpy(siglip_preprocess(s.vit_inputs, size)).to(dev)
            n_vit = len(s.vit_inputs)
        enc = tokenizer([s.prompt], return_tensors="pt", truncation=True, max_length=MAX_PROMPT_TOKENS)["input_ids"]
        return s, pixel_values, n_vit, enc

    if batch_size > 1 and hasattr(model, "generate_stream"):
        # continuous batching: batch_size decode slots over ALL records — a slot freed by a finished record takes the next one at once
        # (no record waits for the slowest member of a fixed group); the rows are the same and in record order.  Only the PLAN of a
        # record (image sizes -> tile counts -> prompt -> token ids: host integers and strings) is made up front; its pixels are decoded,
        # tiled and moved to the device when a slot admits it and dropped after its prefill, so memory does not grow with the shard.
        planned = [plan_record(rec, setting, tokenizer) for rec in records]

        def pixels_of(rec):
            return lambda: prepare(rec)[1]
        kw = generate_kwargs(tokenizer.pad_token_id)
        outs = model.generate_stream([(enc.to(dev), pixels_of(rec)) for rec, (_, _, enc) in zip(records, planned)], batch_size=batch_size,
                                     eos_token_id=kw["eos_token_id"], max_new_tokens=kw["max_new_tokens"], stats=stats)
        for rec, (question, n_vit, enc), out in zip(records, planned, outs):
            response = tokenizer.batch_decode(out[:, enc.shape[1]:], skip_special_tokens=True)[0]
            rows.append(result_row(rec, question, response, n_vit, scorer))
        return rows
    for b0 in range(0, len(records), max(1, batch_size)):
        group = records[b0:b0 + max(1, batch_size)]
        prepared = [prepare(rec) for rec in group]
        kw = generate_kwargs(tokenizer.pad_token_id)
        if len(prepared) == 1 or not hasattr(model, "generate_batch"):
            outs = []
            for s, pixel_values, n_vit, enc in prepared:
                attn = enc != tokenizer.pad_token_id
                outs.append(model.generate(enc.to(dev), pixel_values=pixel_values, attention_mask=attn.to(dev), **kw))
        else:
            outs = model.generate_batch([(enc.to(dev), pixel_values) for _, pixel_values, _, enc in prepared],
                                        eos_token_id=kw["eos_token_id"], max_new_tokens=kw["max_new_tokens"])
        for rec, (s, _, n_vit, enc), out in zip(group, prepared, outs):
            response = tokenizer.batch_decode(out[:, enc.shape[1]:], skip_special_tokens=True)[0]
            rows.append(result_row(rec, s.question, response, n_vit, scorer))
    return rows


def write_jsonl(path: str, rows: List[dict]) -> None:
    with open(path, "w") as f:
        for r in rows:
            f.write(json.dumps(r) + "\n")
