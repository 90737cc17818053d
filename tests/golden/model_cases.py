"""Model-parity cases shared by ``make_golden.py`` (which runs the REFERENCE classes) and the
tests (which run this repo's classes): same seeds, same input generator, same scalar."""
import torch

MODEL_CASES = [  # name, reference class path, class path in this repo, kwargs, recurrent
    ("ff", "rlpyt.models.pg.atari_ff_model.AtariFfModel",
     "rlpyt_amd.models.pg.atari_ff_model.AtariFfModel", dict(), False),
    ("dqn", "rlpyt.models.dqn.atari_dqn_model.AtariDqnModel",
     "rlpyt_amd.models.dqn.atari_dqn_model.AtariDqnModel", dict(), False),
    ("dqn_duel", "rlpyt.models.dqn.atari_dqn_model.AtariDqnModel",
     "rlpyt_amd.models.dqn.atari_dqn_model.AtariDqnModel", dict(dueling=True), False),
    ("cat", "rlpyt.models.dqn.atari_catdqn_model.AtariCatDqnModel",
     "rlpyt_amd.models.dqn.atari_catdqn_model.AtariCatDqnModel", dict(n_atoms=51), False),
    ("cat_duel", "rlpyt.models.dqn.atari_catdqn_model.AtariCatDqnModel",
     "rlpyt_amd.models.dqn.atari_catdqn_model.AtariCatDqnModel",
     dict(n_atoms=51, dueling=True), False),
    ("lstm", "rlpyt.models.pg.atari_lstm_model.AtariLstmModel",
     "rlpyt_amd.models.pg.atari_lstm_model.AtariLstmModel", dict(), True),
    ("r2d1", "rlpyt.models.dqn.atari_r2d1_model.AtariR2d1Model",
     "rlpyt_amd.models.dqn.atari_r2d1_model.AtariR2d1Model", dict(), True),
    ("r2d1_duel", "rlpyt.models.dqn.atari_r2d1_model.AtariR2d1Model",
     "rlpyt_amd.models.dqn.atari_r2d1_model.AtariR2d1Model", dict(dueling=True), True),
]
MODEL_SEED = 1234


def model_inputs(recurrent, A=6, lstm=512):
    """Seeded inputs shared by the generator and the tests (same code on both sides)."""
    g = torch.Generator().manual_seed(99)
    lead = (3, 2) if recurrent else (6,)
    obs = torch.randint(0, 256, lead + (4, 104, 80), dtype=torch.uint8, generator=g)
    act = torch.randint(0, A, lead, generator=g)
    prev_action = torch.nn.functional.one_hot(act, A).float()
    prev_reward = torch.randn(lead, generator=g)
    if not recurrent:
        return obs, prev_action, prev_reward
    h = 0.1 * torch.randn(1, lead[1], lstm, generator=g)
    c = 0.1 * torch.randn(1, lead[1], lstm, generator=g)
    return obs, prev_action, prev_reward, (h, c)


def scalarize(outputs):
    """A fixed scalar of the model outputs whose gradient exercises every parameter."""
    flat = []
    for o in outputs:
        if isinstance(o, torch.Tensor):
            flat.append(o)
        else:  # RnnState
            flat.extend(list(o))
    g = torch.Generator().manual_seed(7)
    return sum((o * torch.randn(o.shape, generator=g).to(o.device)).sum() for o in flat)
