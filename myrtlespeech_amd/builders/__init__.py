"""Protobuf-configured builders: mirror of myrtlespeech/builders/ for the model path.

``build(cfg, ...)`` functions take the messages of ``myrtlespeech_amd.protos`` (same schema
as the reference's ``*_pb2``, so its text-format configs parse unchanged) and return the
MI355X-native modules.  Dataset / optimiser / training-loop builders are out of scope."""
