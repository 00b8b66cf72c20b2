"""Mirror of myrtlespeech/builders/deep_speech_2.py:22-221."""
import math
from typing import Tuple

import torch

from myrtlespeech_amd import protos
from myrtlespeech_amd.builders.activation import build as build_activation
from myrtlespeech_amd.builders.fully_connected import build as build_fully_connected
from myrtlespeech_amd.builders.lookahead import build as build_lookahead
from myrtlespeech_amd.builders.rnn import build as build_rnn
from myrtlespeech_amd.model.cnn import Conv1dTo2d, Conv2dTo1d, MaskConv1d, MaskConv2d, PaddingMode, out_lens
from myrtlespeech_amd.model.deep_speech_2 import DeepSpeech2
from myrtlespeech_amd.model.seq_len_wrapper import SeqLenWrapper


def build(deep_speech_2_cfg, input_features: int, input_channels: int, output_features: int) -> DeepSpeech2:
    """``DeepSpeech2`` for a config: conv blocks -> rnn -> [lookahead] -> fully connected."""
    cnn, cnn_out_features = _build_cnn(deep_speech_2_cfg.conv_block, input_features, input_channels)
    rnn, rnn_out_features = build_rnn(deep_speech_2_cfg.rnn, input_features=cnn_out_features)
    if deep_speech_2_cfg.lookahead_block.HasField("lookahead"):
        lookahead = build_lookahead(deep_speech_2_cfg.lookahead_block.lookahead, input_features=rnn_out_features)
        activation = SeqLenWrapper(build_activation(deep_speech_2_cfg.lookahead_block.activation), torch.nn.Identity())
        # the reference always wraps (its `activation != torch.nn.Identity` compares an instance with a
        # class, builders/deep_speech_2.py:118-123), which fixes the state_dict key `lookahead.0.weight`
        lookahead = torch.nn.Sequential(lookahead, activation)
    else:
        lookahead = None
    fully_connected = build_fully_connected(deep_speech_2_cfg.fully_connected, input_features=rnn_out_features,
                                            output_features=output_features)
    return DeepSpeech2(cnn, rnn, lookahead, fully_connected)


def _padding_mode(value: int) -> PaddingMode:
    if value == protos.PADDING_MODE_NONE:
        return PaddingMode.NONE
    if value == protos.PADDING_MODE_SAME:
        return PaddingMode.SAME
    raise ValueError(f"unknown padding mode {value}")


def _build_cnn(conv_blocks, input_features: int, input_channels: int) -> Tuple[torch.nn.Sequential, int]:
    act_dims = 4  # batch, channels, features, seq_len
    layers = []
    for block in conv_blocks:
        kind = block.WhichOneof("convnd")
        if kind == "conv1d":
            if act_dims == 4:
                layers.append(Conv2dTo1d())
                act_dims = 3
                input_channels *= input_features
                input_features = 1
            cfg = block.conv1d
            layers.append(MaskConv1d(in_channels=input_channels, out_channels=cfg.output_channels,
                                     kernel_size=cfg.kernel_time, stride=cfg.stride_time,
                                     padding_mode=_padding_mode(cfg.padding_mode), bias=cfg.bias))
            input_channels = cfg.output_channels
        elif kind == "conv2d":
            if act_dims == 3:
                layers.append(Conv1dTo2d())
                act_dims = 4
                input_features = input_channels
                input_channels = 1
            cfg = block.conv2d
            mode = _padding_mode(cfg.padding_mode)
            if mode == PaddingMode.NONE:
                input_features = out_lens(torch.tensor([input_features]), kernel_size=cfg.kernel_feature,
                                          stride=cfg.stride_feature, dilation=1, padding=0).item()
            else:
                input_features = math.ceil(input_features / cfg.stride_feature)
            layers.append(MaskConv2d(in_channels=input_channels, out_channels=cfg.output_channels,
                                     kernel_size=[cfg.kernel_feature, cfg.kernel_time],
                                     stride=[cfg.stride_feature, cfg.stride_time], padding_mode=mode, bias=cfg.bias))
            input_channels = cfg.output_channels
        else:
            raise ValueError(f"conv block without a convolution: {block}")
        layers.append(SeqLenWrapper(build_activation(block.activation), torch.nn.Identity()))
    if act_dims == 3:
        layers.append(Conv1dTo2d())
        input_features = input_channels
        input_channels = 1
    return torch.nn.Sequential(*layers), input_features * input_channels
