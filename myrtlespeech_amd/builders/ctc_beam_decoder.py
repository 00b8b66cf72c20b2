"""Mirror of myrtlespeech/builders/ctc_beam_decoder.py:6-80."""
from myrtlespeech_amd.builders.language_model import build as build_lm
from myrtlespeech_amd.post_process.ctc_beam_decoder import CTCBeamDecoder


def build(ctc_beam_decoder_cfg) -> CTCBeamDecoder:
    cfg = ctc_beam_decoder_cfg
    lm = build_lm(cfg.language_model)
    separator_index = None
    if cfg.HasField("separator_index"):
        separator_index = cfg.separator_index.value
        if separator_index == cfg.blank_index:
            raise ValueError(f"separator_index={separator_index} must not be equal to blank_index={cfg.blank_index}")
    lm_weight = cfg.lm_weight.value if cfg.HasField("lm_weight") else None
    return CTCBeamDecoder(blank_index=cfg.blank_index, beam_width=cfg.beam_width, prune_threshold=cfg.prune_threshold,
                          language_model=lm, lm_weight=lm_weight, separator_index=separator_index,
                          word_weight=cfg.word_weight)
