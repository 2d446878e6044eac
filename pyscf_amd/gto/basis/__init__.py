"""Packaged basis-set tables (data.json, extracted by tools/extract_basis.py) and the NWChem-format parser."""
