"""configs/config.json reader shared by the mirrors of the reference modules that read it at
import time (reference: utils/basic_anchors.py:5-9, nets/rpn.py:11-15)."""
import json
import os

_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "configs", "config.json")


def load_config() -> dict:
    with open(_PATH, "r") as f:
        return json.load(f)
