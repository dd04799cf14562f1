"""Keras-compatible layer and op surface of the low-bit path (reference: layers/)."""
