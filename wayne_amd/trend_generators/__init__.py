"""Trend and noise-source generators (mirror of wayne/trend_generators)."""
