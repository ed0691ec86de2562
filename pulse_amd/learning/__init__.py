"""Learner side of the PULSE hot path on MI355X (mirrors phc/learning of the reference)."""
