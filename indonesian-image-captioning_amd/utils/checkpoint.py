"""Checkpoint files with the reference's layout (utils/checkpoint.py:4-31, :34-61): one `torch.save` of a
dict holding the *module and optimizer objects themselves* under the keys 'epoch',
'epochs_since_improvement', 'bleu-4' (or 'accuracy'), 'encoder', 'decoder', 'encoder_optimizer',
'decoder_optimizer', in 'checkpoint_<model>_<data>.pth.tar' (+ a 'BEST_' copy).  Because this overlay keeps
the reference's dotted module paths, such a file unpickles to these classes, and `FusedClampAdam`
(flat parameter/moment buffers) survives the round trip with its parameter views still aliasing the flat
buffer: torch.save keeps storage sharing inside one file.

`load_checkpoint` is the read side of `trains/attention_scn.py:97-111`: whole-object pickles need
`weights_only=False`, so only load files you wrote."""
import os

import torch


def save_checkpoint(model_name, data_name, epoch, epochs_since_improvement, encoder, decoder, encoder_optimizer,
                    decoder_optimizer, bleu4, is_best, folder='.'):
    state = {'epoch': epoch,
             'epochs_since_improvement': epochs_since_improvement,
             'bleu-4': bleu4,
             'encoder': encoder,
             'decoder': decoder,
             'encoder_optimizer': encoder_optimizer,
             'decoder_optimizer': decoder_optimizer}
    filename = 'checkpoint_' + model_name + '_' + data_name + '.pth.tar'
    torch.save(state, os.path.join(folder, filename))
    # the best checkpoint so far is kept separately so a worse one does not overwrite it
    if is_best:
        torch.save(state, os.path.join(folder, 'BEST_' + filename))
    return os.path.join(folder, filename)


def save_tagger_checkpoint(data_name, epoch, epochs_since_improvement, encoder, encoder_optimizer, accuracy, is_best,
                           folder='.'):
    state = {'epoch': epoch,
             'epochs_since_improvement': epochs_since_improvement,
             'accuracy': accuracy,
             'encoder': encoder,
             'encoder_optimizer': encoder_optimizer}
    filename = 'checkpoint_tagger_' + data_name + '.pth.tar'
    torch.save(state, os.path.join(folder, filename))
    if is_best:
        torch.save(state, os.path.join(folder, 'BEST_' + filename))
    return os.path.join(folder, filename)


def load_checkpoint(path, map_location=None):
    return torch.load(path, map_location=map_location, weights_only=False)


def save_state_dicts(filename, encoder=None, decoder=None, tagger=None, **extra):
    """The state-dict style the reference's inference / evaluation tooling reads but none of its scripts writes
    (inference.py:93,118-129, eval_caption.py:65,77-85): `encoder_model_state_dict`, `decoder_model_state_dict`
    for the caption model, `model_state_dict` for the tagger encoder.  Tensors only, so it loads with
    `torch.load(..., weights_only=True)`."""
    state = dict(extra)
    if encoder is not None:
        state['encoder_model_state_dict'] = encoder.state_dict()
    if decoder is not None:
        state['decoder_model_state_dict'] = decoder.state_dict()
    if tagger is not None:
        state['model_state_dict'] = tagger.state_dict()
    torch.save(state, filename)
    return filename
