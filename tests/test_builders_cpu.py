"""Protobuf-configured builders, alphabet and WER harness (8f.2): CPU-side
structure tests restating tests/builders/*, tests/configs/test_configs.py and
tests/data/test_alphabet.py of the reference.  The config texts below restate the values of
the reference's shipped configs (configs/deep_speech_{1,2}_en.config, speech_to_text part)."""
import pytest
import torch

from myrtlespeech_amd import protos as P
from myrtlespeech_amd.builders import activation as b_act
from myrtlespeech_amd.builders import ctc_beam_decoder as b_beam
from myrtlespeech_amd.builders import ctc_loss as b_ctc
from myrtlespeech_amd.builders import fully_connected as b_fc
from myrtlespeech_amd.builders import language_model as b_lm
from myrtlespeech_amd.builders import rnn as b_rnn
from myrtlespeech_amd.builders.speech_to_text import build as build_stt
from myrtlespeech_amd.data.alphabet import Alphabet
from myrtlespeech_amd.stage import Stage
from myrtlespeech_amd.wer import WordErrorRate, WordSegmentor

DS1_EN = '''
alphabet: " abcdefghijklmnopqrstuvwxyz'_";
pre_process_step { stage: TRAIN_AND_EVAL; mfcc { n_mfcc: 26; win_length: 400; hop_length: 320; } }
pre_process_step { stage: TRAIN; spec_augment { feature_mask: 3; time_mask: 20; n_feature_masks: 2; n_time_masks: 2; } }
pre_process_step { stage: TRAIN_AND_EVAL; context_frames { n_context: 9; } }
deep_speech_1 { n_hidden: 1024; drop_prob: 0.25; relu_clip: 20.0; forget_gate_bias: 1.0; }
ctc_loss { blank_index: 28; reduction: SUM; }
ctc_greedy_decoder { blank_index: 28; }
'''

DS2_EN = '''
alphabet: " abcdefghijklmnopqrstuvwxyz'_";
pre_process_step { stage: TRAIN_AND_EVAL; mfcc { n_mfcc: 80; win_length: 400; hop_length: 160; } }
pre_process_step { stage: TRAIN_AND_EVAL; standardize { } }
deep_speech_2 {
  conv_block {
    conv2d { output_channels: 32; kernel_feature: 41; kernel_time: 11; stride_feature: 2; stride_time: 2;
             padding_mode: SAME; bias: true; }
    activation { hardtanh { min_val: 0.0; max_val: 20.0; } }
  }
  conv_block {
    conv2d { output_channels: 32; kernel_feature: 21; kernel_time: 11; stride_feature: 2; stride_time: 1;
             padding_mode: SAME; bias: true; }
    activation { hardtanh { min_val: 0.0; max_val: 20.0; } }
  }
  rnn { rnn_type: GRU; hidden_size: 2560; num_layers: 3; bias: true; bidirectional: false; }
  lookahead_block { lookahead { context: 80; } activation { identity {} } }
  fully_connected { num_hidden_layers: 1; hidden_size: 1024; activation { hardtanh { min_val: 0.0; max_val: 20.0; } } }
}
ctc_loss { blank_index: 28; reduction: SUM; }
ctc_greedy_decoder { blank_index: 28; }
'''


def test_shipped_style_configs_build():
    """tests/configs/test_configs.py:37-56: every shipped config builds."""
    ds1 = build_stt(P.parse(DS1_EN, P.SpeechToText))
    assert ds1.model.__class__.__name__ == "DeepSpeech1"
    assert ds1.model.fc1[0].in_features == 26 * 19 and ds1.model.out.out_features == 29
    assert [s for _, s in ds1.pre_process_steps] == [Stage.TRAIN_AND_EVAL, Stage.TRAIN, Stage.TRAIN_AND_EVAL]
    ds2 = build_stt(P.parse(DS2_EN, P.SpeechToText))
    m = ds2.model
    assert m.rnn.rnn.__class__ is torch.nn.GRU and m.rnn.rnn.input_size == 32 * 20 and m.rnn.rnn.hidden_size == 2560
    assert "lookahead.0.weight" in m.state_dict()
    assert m.lookahead[0].in_features == 2560 and m.lookahead[0].context == 80
    assert m.fully_connected.fully_connected[0].in_features == 2560
    assert ds2.post_process.blank_index == 28 and len(ds2.alphabet) == 29
    assert ds2.loss.ctc_loss.reduction == "sum"
    r = repr(m.cnn[0])
    assert r == "MaskConv2d(1, 32, kernel_size=(41, 11), stride=(2, 2), padding_mode=PaddingMode.SAME)"


def test_conv1d_blocks_and_no_lookahead():
    cfg = P.parse('''
    alphabet: "ab_";
    pre_process_step { stage: TRAIN_AND_EVAL; mfcc { n_mfcc: 12; win_length: 400; hop_length: 160; } }
    deep_speech_2 {
      conv_block { conv2d { output_channels: 3; kernel_feature: 5; kernel_time: 3; stride_feature: 2; stride_time: 2;
                            padding_mode: SAME; bias: true; } activation { relu {} } }
      conv_block { conv1d { output_channels: 10; kernel_time: 3; stride_time: 1; padding_mode: NONE; bias: false; }
                   activation { identity {} } }
      rnn { rnn_type: LSTM; hidden_size: 8; num_layers: 2; bias: true; bidirectional: true; forget_gate_bias { value: 1.0 } }
      lookahead_block { no_lookahead {} activation { identity {} } }
      fully_connected { num_hidden_layers: 0; activation { identity {} } }
    }
    ctc_loss { blank_index: 2; reduction: MEAN; }
    ctc_beam_decoder { blank_index: 2; beam_width: 4; prune_threshold: 0.01; language_model { no_lm {} }
                       separator_index { value: 0 } word_weight: 1.5 }
    ''', P.SpeechToText)
    stt = build_stt(cfg)
    names = [l.__class__.__name__ for l in stt.model.cnn]
    assert names == ["MaskConv2d", "SeqLenWrapper", "Conv2dTo1d", "MaskConv1d", "SeqLenWrapper", "Conv1dTo2d"]
    assert stt.model.cnn[3].in_channels == 3 * 6 and stt.model.cnn[3].bias is None
    assert stt.model.rnn.rnn.input_size == 10 and stt.model.lookahead is None
    assert isinstance(stt.model.fully_connected.fully_connected, torch.nn.Linear)
    assert stt.post_process.beam_width == 4 and stt.post_process.separator_index == 0
    assert abs(stt.post_process.prune_threshold - 0.01) < 1e-7 and stt.post_process.language_model is None


def test_builder_value_errors():
    base = P.parse(DS1_EN, P.SpeechToText)
    bad = P.SpeechToText()
    bad.CopyFrom(base)
    bad.ctc_greedy_decoder.blank_index = 3                       # mismatch with ctc_loss (speech_to_text.py:232-233)
    with pytest.raises(ValueError):
        build_stt(bad)
    bad.CopyFrom(base)
    bad.ctc_loss.blank_index = 40                                # out of range (speech_to_text.py:192-199)
    with pytest.raises(ValueError):
        build_stt(bad)
    bad.CopyFrom(base)
    bad.ClearField("deep_speech_1")                              # no model
    with pytest.raises(ValueError):
        build_stt(bad)
    beam = P.parse("blank_index: 1; beam_width: 2; language_model { no_lm {} } separator_index { value: 1 }",
                   P.CTCBeamDecoder)
    with pytest.raises(ValueError):                              # separator == blank (ctc_beam_decoder.py:52-56)
        b_beam.build(beam)
    with pytest.raises(ValueError):
        b_lm.build(P.LanguageModel())                            # nothing set
    with pytest.raises(ValueError):
        b_act.build(P.Activation())
    assert b_lm.build(P.parse("no_lm {}", P.LanguageModel)) is None


def test_small_builders():
    act = b_act.build(P.parse("hardtanh { min_val: 0.0; max_val: 20.0 }", P.Activation))
    assert repr(act) == "Hardtanh(min_val=0.0, max_val=20.0)"
    assert isinstance(b_act.build(P.parse("relu {}", P.Activation)), torch.nn.ReLU)
    rnn, out = b_rnn.build(P.parse("rnn_type: BASIC_RNN; hidden_size: 5; num_layers: 2; bias: true; bidirectional: true;",
                                   P.RNN), input_features=7)
    assert out == 10 and rnn.rnn.__class__ is torch.nn.RNN and rnn.rnn.num_layers == 2
    fc = b_fc.build(P.parse("num_hidden_layers: 2; hidden_size: 64; activation { relu {} }", P.FullyConnected), 32, 16)
    got = [m.__class__.__name__ for m in fc.fully_connected]
    assert got == ["Linear", "ReLU", "Linear", "ReLU", "Linear"] and fc.fully_connected[4].out_features == 16
    loss = b_ctc.build(P.parse("blank_index: 0; reduction: SUM;", P.CTCLoss))
    assert loss.ctc_loss.blank == 0 and loss.ctc_loss.reduction == "sum"


def test_alphabet():
    a = Alphabet(["a", "b", "c", ".", " "])
    assert repr(a) == "Alphabet(symbols=['a', 'b', 'c', '.', ' '])" and len(a) == 5
    assert a[1] == "b" and a.get_symbol(9) is None and a.get_index("z") is None
    assert a.get_symbols([0, 7, 2]) == ["a", "c"] and a.get_indices(list("a?c ")) == [0, 2, 4]
    with pytest.raises(IndexError):
        a[5]
    with pytest.raises(ValueError):
        Alphabet(["a", "a"])


def test_word_error_rate():
    a = Alphabet(list(" abc"))
    seg = WordSegmentor(" ")
    assert seg(list("  ab  c a ")) == ["ab", "c", "a"]
    w = WordErrorRate(a, seg)
    hyp = [a.get_indices(list("ab c")), a.get_indices(list("a"))]
    tgt = torch.tensor([a.get_indices(list("ab ca")) + [0, 0], a.get_indices(list("a b c")) + [0] * 2])
    w.update(hyp, tgt, torch.tensor([5, 5]))
    # "ab c" vs "ab ca": 1 substitution of 2 words; "a" vs "a b c": 2 deletions of 3 words -> 3/5
    assert w.distances == [1, 2] and w.lengths == [2, 3] and abs(w.value() - 60.0) < 1e-9


# ----------------------------------------------------------------------------- the reference's own shipped configs
REF_CONFIGS = "/root/reference/src/myrtlespeech/configs"


def _speech_to_text_block(text: str) -> str:
    """The body of the ``speech_to_text { ... }`` message of a TaskConfig text file (the train / eval / dataset parts of
    the TaskConfig are the control plane, out of scope: tests/configs/test_configs.py:37-56 swaps them for fakes too)."""
    start = text.index("speech_to_text")
    i = text.index("{", start)
    depth, j = 0, i
    in_str = False
    while True:
        c = text[j]
        if c == '"' and text[j - 1] != "\\":
            in_str = not in_str
        elif not in_str:
            depth += c == "{"
            depth -= c == "}"
            if depth == 0:
                return text[i + 1:j]
        j += 1


@pytest.mark.skipif(not __import__("os").path.isdir(REF_CONFIGS), reason="reference checkout not mounted (GPU box)")
@pytest.mark.parametrize("name", ["deep_speech_1_en.config", "deep_speech_2_en.config"])
def test_reference_shipped_config_files_parse_and_build(name):
    """tests/configs/test_configs.py:37-56 on the REAL files: the speech_to_text block of every shipped .config parses
    with the runtime descriptors and builds onto the accelerated modules (read as data; nothing is copied)."""
    import os
    with open(os.path.join(REF_CONFIGS, name)) as f:
        cfg = P.parse(_speech_to_text_block(f.read()), P.SpeechToText)
    stt = build_stt(cfg)
    n_params = sum(p.numel() for p in stt.model.parameters())
    if name.startswith("deep_speech_1"):
        assert stt.model.__class__.__name__ == "DeepSpeech1" and 30.9e6 < n_params < 31.1e6
    else:
        assert stt.model.__class__.__name__ == "DeepSpeech2" and 106e6 < n_params < 107e6
        assert stt.model.rnn.rnn.__class__ is torch.nn.GRU and stt.model.rnn.rnn.hidden_size == 2560
        assert "lookahead.0.weight" in stt.model.state_dict()
    assert stt.post_process.blank_index == 28 and len(stt.alphabet) == 29
    # and the restated texts above say the same thing as the files
    restated = build_stt(P.parse(DS1_EN if name.startswith("deep_speech_1") else DS2_EN, P.SpeechToText))
    assert [(k, tuple(v.shape)) for k, v in restated.model.state_dict().items()] == \
           [(k, tuple(v.shape)) for k, v in stt.model.state_dict().items()]
    assert [s for _, s in restated.pre_process_steps] == [s for _, s in stt.pre_process_steps]
