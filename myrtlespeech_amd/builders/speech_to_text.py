"""Mirror of myrtlespeech/builders/speech_to_text.py:29-272: alphabet, pre-processing steps (device front-end,
``builders/pre_process_step.py``), model, loss and decoder for a ``SpeechToText`` config, with the reference's
``ValueError`` checks on blank / separator indices."""
from typing import Callable, List, Tuple

from myrtlespeech_amd.builders.ctc_beam_decoder import build as build_ctc_beam_decoder
from myrtlespeech_amd.builders.ctc_loss import build as build_ctc_loss
from myrtlespeech_amd.builders.deep_speech_2 import build as build_deep_speech_2
from myrtlespeech_amd.builders.pre_process_step import build as build_pre_process_step
from myrtlespeech_amd.data.alphabet import Alphabet
from myrtlespeech_amd.data.preprocess import AddContextFrames, MFCC, MFCCLegacy, SpecAugment, Standardize
from myrtlespeech_amd.model.cnn import Conv1dTo2d
from myrtlespeech_amd.model.deep_speech_1 import DeepSpeech1
from myrtlespeech_amd.model.speech_to_text import SpeechToText
from myrtlespeech_amd.post_process.ctc_greedy_decoder import CTCGreedyDecoder
from myrtlespeech_amd.stage import Stage


def _check_index(name: str, value: int, alphabet: Alphabet) -> None:
    hi = max(0, len(alphabet) - 1)
    if not 0 <= value <= hi:
        raise ValueError(f"{name}={value} must be in [0, {hi}]")


def build(stt_cfg) -> SpeechToText:
    """``SpeechToText`` (model + loss + decoder + alphabet) for a ``SpeechToText`` config."""
    alphabet = Alphabet(list(stt_cfg.alphabet))
    pre_process_steps, input_features, input_channels = _build_pre_process_steps(stt_cfg.pre_process_step)

    model_type = stt_cfg.WhichOneof("supported_models")
    if model_type == "deep_speech_1":
        c = stt_cfg.deep_speech_1
        model = DeepSpeech1(input_features=input_features, input_channels=input_channels, n_hidden=c.n_hidden,
                            out_features=len(alphabet), drop_prob=c.drop_prob, relu_clip=c.relu_clip,
                            forget_gate_bias=c.forget_gate_bias, hard_lstm=c.hard_lstm)
    elif model_type == "deep_speech_2":
        model = build_deep_speech_2(stt_cfg.deep_speech_2, input_features=input_features,
                                    input_channels=input_channels, output_features=len(alphabet))
    else:
        raise ValueError(f"model={model_type} not supported")

    blank_indices: List[int] = []
    loss_type = stt_cfg.WhichOneof("supported_losses")
    if loss_type == "ctc_loss":
        blank_indices.append(stt_cfg.ctc_loss.blank_index)
        _check_index("ctc_loss.blank_index", stt_cfg.ctc_loss.blank_index, alphabet)
        loss = build_ctc_loss(stt_cfg.ctc_loss)
    else:
        raise ValueError(f"loss={loss_type} not supported")

    post_type = stt_cfg.WhichOneof("supported_post_processes")
    if post_type == "ctc_greedy_decoder":
        blank_indices.append(stt_cfg.ctc_greedy_decoder.blank_index)
        _check_index("ctc_greedy_decoder.blank_index", stt_cfg.ctc_greedy_decoder.blank_index, alphabet)
        post_process = CTCGreedyDecoder(blank_index=stt_cfg.ctc_greedy_decoder.blank_index)
    elif post_type == "ctc_beam_decoder":
        blank_indices.append(stt_cfg.ctc_beam_decoder.blank_index)
        _check_index("ctc_beam_decoder.blank_index", stt_cfg.ctc_beam_decoder.blank_index, alphabet)
        if stt_cfg.ctc_beam_decoder.HasField("separator_index"):
            _check_index("ctc_beam_decoder.separator_index.value", stt_cfg.ctc_beam_decoder.separator_index.value,
                         alphabet)
        post_process = build_ctc_beam_decoder(stt_cfg.ctc_beam_decoder)
    else:
        raise ValueError(f"post_process={post_type} not supported")

    if blank_indices and len(set(blank_indices)) != 1:
        raise ValueError("all blank_index values of CTC components must match")
    return SpeechToText(alphabet=alphabet, model=model, loss=loss, pre_process_steps=pre_process_steps,
                        post_process=post_process)


def _build_pre_process_steps(step_cfgs) -> Tuple[List[Tuple[Callable, Stage]], int, int]:
    """(steps, input_features, input_channels): an MFCC step fixes the feature count, context frames the channel
    count; a config without an MFCC step feeds raw ``[N, 1, T]`` audio through ``Conv1dTo2d`` (one feature)."""
    input_features = None
    input_channels = 1
    steps: List[Tuple[Callable, Stage]] = []
    for cfg in step_cfgs:
        step, stage = build_pre_process_step(cfg)
        if isinstance(step, (MFCC, MFCCLegacy)):
            input_features = step.n_mfcc
        elif isinstance(step, AddContextFrames):
            input_channels = 2 * step.n_context + 1
        elif not isinstance(step, (SpecAugment, Standardize)):
            raise ValueError(f"unknown step={step}")
        steps.append((step, stage))
    if input_features is None:
        steps.append((Conv1dTo2d(seq_len_support=False), Stage.TRAIN_AND_EVAL))
        input_features = 1
    return steps, input_features, input_channels
