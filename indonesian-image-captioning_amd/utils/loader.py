"""`load_decoder` with the reference's signature (utils/loader.py:9-68): build the decoder named by
`model_type` and load a `state_dict` into it.  The device is picked when called, not at import."""
from utils.device import get_device

scn_based_model = {'pure_scn', 'attention_scn'}
att_based_model = {'pure_attention', 'attention_scn'}


def load_decoder(model_type, checkpoint, vocab_size, embed_dim=512, attention_dim=512, decoder_dim=512,
                 factored_dim=512, semantic_dim=1000, dropout=0.5):
    if model_type == 'pure_scn':
        from models.decoders.pure_scn import PureSCN
        decoder_caption = PureSCN(embed_dim=embed_dim, decoder_dim=decoder_dim, factored_dim=factored_dim,
                                  semantic_dim=semantic_dim, vocab_size=vocab_size, dropout=dropout)
    elif model_type == 'pure_attention':
        from models.decoders.pure_attention import PureAttention
        decoder_caption = PureAttention(attention_dim=attention_dim, embed_dim=embed_dim, decoder_dim=decoder_dim,
                                        vocab_size=vocab_size, dropout=dropout)
    elif model_type == 'attention_scn':
        from models.decoders.attention_scn import AttentionSCN
        # the reference hard-codes semantic_dim=1000 here (loader.py:58); honour the argument instead
        decoder_caption = AttentionSCN(attention_dim=attention_dim, embed_dim=embed_dim, decoder_dim=decoder_dim,
                                       factored_dim=factored_dim, semantic_dim=semantic_dim, vocab_size=vocab_size,
                                       dropout=dropout)
    else:
        raise ValueError('Error model type not found!')
    decoder_caption.load_state_dict(checkpoint)
    return decoder_caption.to(get_device())
