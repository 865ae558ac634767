"""Plain-dict restatement of the reference's sacred config (vilt/config.py:24-116) with the
``task_moco`` overrides (:128-164).  sacred itself is out of scope; the KEYS and DEFAULTS are part
of the drop-in boundary."""


def _loss_names(d):
    ret = {"moco": 0, "barlowtwins": 0, "itm": 0, "mlm": 0, "mpp": 0, "vqa": 0, "nlvr2": 0, "irtr": 0,
           "irtr_attacked": 0, "nlvr2_attacked": 0, "vqa_attacked": 0}
    ret.update(d)
    return ret


def default_config(**over):
    cfg = dict(
        exp_name="vilt", seed=0, loss_names=_loss_names({"itm": 1, "mlm": 1}), batch_size=4096,
        image_size=384, max_image_len=-1, patch_size=32, draw_false_image=1, image_only=False,
        vqav2_label_size=3129, max_text_len=40, tokenizer="bert-base-uncased", vocab_size=30522,
        whole_word_masking=False, mlm_prob=0.15, draw_false_text=0,
        vit="vit_base_patch32_384", hidden_size=768, num_heads=12, num_layers=12, mlp_ratio=4, drop_rate=0.1,
        optim_type="adamw", learning_rate=1e-4, weight_decay=0.01, decay_power=1, max_epoch=100, max_steps=25000,
        warmup_steps=2500, end_lr=0, lr_mult=1,
        get_recall_metric=False, resume_from="", fast_dev_run=False, val_check_interval=1.0, test_only=False,
        data_root="", log_dir="result", per_gpu_batchsize=0, num_gpus=1, num_nodes=1, load_path="",
        num_workers=8, precision=16,
    )
    cfg.update(over)
    return cfg


def task_moco(**over):
    cfg = default_config(
        exp_name="moco", Multimodal=True, num_negative=65536, momentum=0.999, temperature=0.07,
        augmentation=False, text_view=False, image_view=False, loss_names=_loss_names({"moco": 1}),
        batch_size=128, max_epoch=1, max_image_len=200, test_only=False,
        adv_steps_img=5, adv_lr_img=0.05, adv_max_norm_img=0.005,
        n_candidates=5, max_loops=10, sim_thred=0.5, cos_sim=True, synonym="cos_sim",
        # reference config.py:161-162; the word-level text attack is active when `tokenizer` is a LOCAL vocab path (or an
        # object) and embedding_path exists - with the hub name "bert-base-uncased" it runs at token level
        embedding_path="../attack/counter-fitted-vectors.txt", sim_path="../attack/cos_sim_counter_fitting.npy", stopwords=None,
        TSNE_vizualisation=False, img_save_path="",
    )
    cfg.update(over)
    return cfg


def task_barlowtwins(**over):
    """reference config.py:166-199: the Barlow-Twins variant of the contrastive pre-training.
    barlowtwins_dims: widths of BarlowTwinsHead - the reference hard-codes [8192, 8192], 8192 (vilt_module.py:115)."""
    cfg = default_config(
        exp_name="barlowtwins", Multimodal=True, augmentation=False, text_view=False, image_view=False,
        loss_names=_loss_names({"barlowtwins": 1}), adv_lr=0.0051, batch_size=128, max_epoch=1, max_image_len=200, test_only=False,
        adv_steps_img=5, adv_lr_img=0.05, adv_max_norm_img=0.005,
        n_candidates=5, max_loops=10, sim_thred=0.5, cos_sim=True, synonym="cos_sim",
        embedding_path="../attack/counter-fitted-vectors.txt", sim_path="../attack/cos_sim_counter_fitting.npy", stopwords=None,
        TSNE_vizualisation=False, img_save_path="", barlowtwins_dims=(8192, 8192, 8192),
    )
    cfg.update(over)
    return cfg
