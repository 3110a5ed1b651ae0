"""Special tokens of the word map (mirror of the reference's utils/token.py:1-4)."""
start_token, end_token, unknown_token, padding_token = '<start>', '<end>', '<unk>', '<pad>'
