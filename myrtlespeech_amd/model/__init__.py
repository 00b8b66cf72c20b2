"""Encoder modules: masked convolutions, recurrent layers, DS1/DS2 compositions, RNN-T parts."""
