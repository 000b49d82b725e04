"""`python -m loco_hd` (/root/reference/loco_hd/__main__.py:149-204): the command line of loco_hd_amd."""
import sys

from loco_hd_amd.__main__ import main

if __name__ == "__main__":
    sys.exit(main())
