"""CTC loss."""
