"""ORACLE -- test infrastructure only.  See oracle/tph_ref.py for the header that applies to the whole directory."""
