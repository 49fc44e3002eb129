"""Sentinel ids and special strings of the prompt format (host side of the kept API).

Mirrors the values of revisionllm/constants.py:7-15 - they are part of the on-the-wire contract between
``tokenizer_image_token`` and the splice step, so they must be identical.
"""
IGNORE_INDEX = -100
IMAGE_TOKEN_INDEX = -200
MEMORY_TOKEN_INDEX = -300
DEFAULT_IMAGE_TOKEN = "<video>"
DEFAULT_MEMORY_TOKEN = "<memory>"
DEFAULT_IGNORE_TOKEN = "<ignore>"

#: memory prefixes (revisionllm/constants.py:14-15), the streaming-memory variant's prefix texts (tokenised by the caller into ``prefix_memory``)
PREFIX = ["Here is an example of a past memory where the event did not occur: ",
          "Here is an example of a past memory where the event did take place: "]
