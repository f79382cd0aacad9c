"""Checkpoint interchange with the reference trainer (trains/trainer.py:475-539):
`torch.load(path)['state_dict']`, optional 'module.' prefix from DataParallel, shape mismatch
or missing keys keep the model's own initialisation (strict=False), as the reference does."""
import torch


def load_model(model, model_path):
    ck = torch.load(model_path, map_location="cpu", weights_only=False)
    print("loaded {}, epoch {}".format(model_path, ck.get("epoch", "?")))
    src = ck["state_dict"] if "state_dict" in ck else ck
    sd = {(k[7:] if k.startswith("module") and not k.startswith("module_list") else k): v for k, v in src.items()}
    own = model.state_dict()
    for k in list(sd):
        if k in own:
            if tuple(sd[k].shape) != tuple(own[k].shape):
                print("Skip loading parameter {}, required shape{}, loaded shape{}.".format(
                    k, tuple(own[k].shape), tuple(sd[k].shape)))
                sd[k] = own[k]
        else:
            print("Drop parameter {}.".format(k))
            del sd[k]
    for k in own:
        if k not in sd:
            print("No param {}.".format(k))
            sd[k] = own[k]
    model.load_state_dict(sd, strict=False)
    return model


def save_model(path, epoch, model):
    torch.save({"epoch": epoch, "state_dict": model.state_dict()}, path)
