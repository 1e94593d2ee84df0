"""common.quantity -- drop-in for the reference package of the same import path
(reference: quantity/common/quantity/__init__.py:1-6 exports the same 21 names)."""
