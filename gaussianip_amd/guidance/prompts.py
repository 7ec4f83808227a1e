"""View-dependent prompt lookup of the AHDS step.

Reference: threestudio/models/prompt_processors/base.py — `PromptProcessorOutput.get_text_embeddings` :52-81, the 13
`DirectionConfig`s of the shipped configuration (`view_dependent_prompt_front: false`) :284-333, `direction2idx` :335,
`prompts_vd` / `negative_prompts_vd` :366-375, `PromptProcessor.__call__` :537-551.  Pinned by
tests/golden/prompt_directions.npz (captured from the reference's own classes).

Quirk reproduced on purpose: the 13 directions re-use six names twice and `direction2idx` is keyed by
NAME, so a name resolves to the index of its second ("full body photo, ...") entry: whatever `all_vis_all` says, the
partial-visibility prompts 0-5 are only ever selected as the default index 0 ("left front", when no window matches).

The text encoder (CLIP tokenizer + text model of the base checkpoint) is not part of this path and cannot ship:
`PromptProcessor` takes an `encode(list of str) -> [n, 77, 768]` callable and builds the tables once.
"""
from dataclasses import dataclass
from typing import Callable, Dict, List, Sequence, Tuple

import torch


@dataclass(frozen=True)
class Direction:
    name: str
    template: str          # "{}" is the base prompt
    all_vis: int           # the all_vis_all value the condition asks for (ignored for "overhead")
    lo: float              # azimuth window, degrees, both ends exclusive (strict comparisons as the reference)
    hi: float


_WINDOWS = (("left front", 0.0, 45.0), ("left back", -45.0, 0.0), ("right front", 135.0, float("inf")),
            ("right back", float("-inf"), -135.0), ("front", 45.0, 135.0), ("back", -135.0, -45.0))
DIRECTIONS: Tuple[Direction, ...] = tuple(
    [Direction(n, "{}, %s view" % n, 0, lo, hi) for n, lo, hi in _WINDOWS] +
    [Direction(n, "{}, full body photo, %s view" % n, 1, lo, hi) for n, lo, hi in _WINDOWS] +
    [Direction("overhead", "{}, overhead view", -1, 0.0, float("inf"))])
# dict keyed by name: later entries overwrite earlier ones, exactly as base.py:335
DIRECTION2IDX: Dict[str, int] = {d.name: i for i, d in enumerate(DIRECTIONS)}


_TABLES = {}


def _tables(device):
    """Window bounds / visibility flags of the 12 azimuth directions and the name-keyed target index of all 13, as
    device tensors (built once per device)."""
    t = _TABLES.get(device)
    if t is None:
        win = [d for d in DIRECTIONS if d.name != "overhead"]
        t = _TABLES[device] = dict(
            lo=torch.tensor([d.lo for d in win], dtype=torch.float32, device=device),
            hi=torch.tensor([d.hi for d in win], dtype=torch.float32, device=device),
            vis=torch.tensor([float(d.all_vis) for d in win], dtype=torch.float32, device=device),
            order=torch.arange(1, len(DIRECTIONS) + 1, device=device),
            target=torch.tensor([0] + [DIRECTION2IDX[d.name] for d in DIRECTIONS], dtype=torch.long, device=device))
    return t


def direction_index(elevation, azimuth, center, all_vis_all, camera_distances=None, head_offset=0.65):
    """Per-view index into the [13, 77, 768] tables (base.py:66-68): the conditions are applied in list order, each
    writing direction2idx[name]; "overhead" = (center == head_offset) & (azimuth > 0) comes last and wins; views that
    match nothing keep index 0.  Evaluated as one [B, 13] condition matrix (a dozen small kernels, no masked
    assignment, no host synchronisation) — the LAST matching direction decides, like the reference's loop."""
    t = _tables(azimuth.device)
    az = azimuth.to(torch.float32)[:, None]
    cond = (all_vis_all.to(torch.float32)[:, None] == t["vis"]) & (az > t["lo"]) & (az < t["hi"])          # [B, 12]
    over = (center.to(az.device) == head_offset) & (azimuth > 0)
    cond = torch.cat([cond, over[:, None]], dim=1)                                                        # [B, 13]
    last = (cond.to(torch.long) * t["order"]).amax(dim=1)                  # 1-based position of the last match, 0 = none
    return t["target"][last]


def view_dependent_prompts(prompt: str) -> List[str]:
    return [d.template.format(prompt) for d in DIRECTIONS]


@dataclass
class PromptProcessorOutput:
    """Same fields and method as the reference's dataclass (the perp-neg members are carried, unused, like there)."""
    text_embeddings: torch.Tensor               # [1, 77, 768]
    uncond_text_embeddings: torch.Tensor        # [1, 77, 768]
    null_embeddings: torch.Tensor               # [1, 77, 768]
    text_embeddings_vd: torch.Tensor            # [13, 77, 768]
    uncond_text_embeddings_vd: torch.Tensor     # [13, 77, 768]
    directions: Sequence[Direction] = DIRECTIONS
    direction2idx: Dict[str, int] = None
    use_perp_neg: bool = False
    perp_neg_f_sb: Tuple[float, float, float] = (1, 0.5, -0.606)
    perp_neg_f_fsb: Tuple[float, float, float] = (1, 0.5, +0.967)
    perp_neg_f_fs: Tuple[float, float, float] = (4, 0.5, -2.426)
    perp_neg_f_sf: Tuple[float, float, float] = (4, 0.5, -2.426)
    head_offset: float = 0.65

    def get_text_embeddings(self, elevation, azimuth, center, all_vis_all, camera_distances, view_dependent_prompting=True):
        """cat[cond, uncond (negative), null] = [3B, 77, 768] — the reference's order, "different from other
        implementations" (base.py:80)."""
        B = elevation.shape[0]
        if view_dependent_prompting:
            idx = direction_index(elevation, azimuth, center, all_vis_all, camera_distances, self.head_offset)
            idx = idx.to(self.text_embeddings_vd.device, non_blocking=True)      # a host-side batch: the lookup ran on the host
            text, uncond = self.text_embeddings_vd[idx], self.uncond_text_embeddings_vd[idx]
        else:
            text = self.text_embeddings.expand(B, -1, -1)
            uncond = self.uncond_text_embeddings.expand(B, -1, -1)
        null = self.null_embeddings.expand(B, -1, -1)
        return torch.cat([text, uncond, null], dim=0)


class PromptProcessor:
    """Host-side counterpart of `PromptProcessor` (base.py:169-551) for the shipped configuration: builds the 13
    view-dependent prompts, encodes [prompt, negative, 13 x vd, 13 x negative vd, ""] once with the caller's text
    encoder and returns the tables on call.  Attributes `prompt`, `negative_prompt`, `null_prompt` are what
    GaussianIP.on_fit_start hands to guidance.prepare_for_sds (GaussianIP.py:355-356)."""

    def __init__(self, prompt: str, encode: Callable[[List[str]], torch.Tensor], negative_prompt: str = "",
                 null_prompt: str = "", head_offset: float = 0.65):
        self.prompt, self.negative_prompt, self.null_prompt = prompt, negative_prompt, null_prompt
        self.head_offset = head_offset
        self.directions = DIRECTIONS
        self.direction2idx = dict(DIRECTION2IDX)
        self.prompts_vd = view_dependent_prompts(prompt)
        self.negative_prompts_vd = [negative_prompt for _ in DIRECTIONS]        # negative_prompt lambdas are identities
        # base.py:385-391: the last entry is the literal empty prompt, whatever cfg.null_prompt says (it is "" in the
        # shipped configuration); load_text_embeddings reads the null embedding from that entry (:437)
        texts = [prompt, negative_prompt] + self.prompts_vd + self.negative_prompts_vd + [""]
        unique = list(dict.fromkeys(texts))                                      # the reference caches by prompt hash
        table = dict(zip(unique, encode(unique)))
        self.text_embeddings = table[prompt][None]
        self.uncond_text_embeddings = table[negative_prompt][None]
        self.text_embeddings_vd = torch.stack([table[p] for p in self.prompts_vd])
        self.uncond_text_embeddings_vd = torch.stack([table[p] for p in self.negative_prompts_vd])
        self.null_embeddings = table[""][None]

    def __call__(self) -> PromptProcessorOutput:
        return PromptProcessorOutput(self.text_embeddings, self.uncond_text_embeddings, self.null_embeddings,
                                     self.text_embeddings_vd, self.uncond_text_embeddings_vd, self.directions,
                                     self.direction2idx, head_offset=self.head_offset)
