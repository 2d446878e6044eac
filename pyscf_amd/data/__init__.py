"""Deterministic benchmark geometries (water clusters of SURVEY.md 8d, benzene)."""
