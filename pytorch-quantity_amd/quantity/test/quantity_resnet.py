"""Alias for the script name the reference's documentation uses: README.md:32 (and BASELINE.json config 1)
say `python quantity_resnet.py`; the file that ships there is quantity/test/resnet18_quantity.py.  Run from
this directory (cwd-relative configs), exactly like that script."""
from resnet18_quantity import main

if __name__ == "__main__":
    main()
