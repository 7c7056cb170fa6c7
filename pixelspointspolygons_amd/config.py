"""Plain attribute-dict config carrying exactly the keys the reference models read (SURVEY §5, config/**/*.yaml).

Hydra / OmegaConf are not required: `make_config()` restates the shipped defaults of
config/encoder/{vit,pointpillars_vit,early_fusion_vit,early_fusion_vit_cnn}.yaml + config/model/{pix2poly,ffl}.yaml.
An OmegaConf DictConfig produced by the reference's own scripts works as well (same attribute access).
"""


class AttrDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    @staticmethod
    def wrap(d):
        if isinstance(d, dict):
            return AttrDict({k: AttrDict.wrap(v) for k, v in d.items()})
        return d

    def values_list(self):
        return list(self.values())


_ENCODERS = {
    "vit": dict(use_images=True, use_lidar=False),
    "pointpillars_vit": dict(use_images=False, use_lidar=True),
    "early_fusion_vit": dict(use_images=True, use_lidar=True),
    "vit_cnn": dict(use_images=True, use_lidar=False),
    "pointpillars_vit_cnn": dict(use_images=False, use_lidar=True),
    "early_fusion_vit_cnn": dict(use_images=True, use_lidar=True),
}


def make_config(encoder="early_fusion_vit", model="pix2poly", *, in_size=224, patch_size=8, patch_feature_dim=384,
                vit_depth=12, vit_heads=6, vit_mlp=None, max_num_vertices=192, out_feature_dim=256,
                max_num_points_per_voxel=64, sinkhorn_iterations=100, device="cuda", multi_gpu=False,
                lidar_dropout=None, precision="bf16", batch_size=16):
    """precision: 'bf16' (bf16 storage, fp32 accumulate; throughput mode), 'fp32' (exact fp32 MFMA; parity mode) or 'fp32x3' (fp32 storage, GEMMs as bf16 x 3 on
    the bf16 MFMA: 2^-17 per product - the north star's 1e-3 at a multiple of the exact mode's throughput)."""
    if encoder not in _ENCODERS:
        raise NotImplementedError(f"Encoder {encoder} not implemented")
    g = in_size // patch_size
    enc = dict(
        name=encoder, **_ENCODERS[encoder], in_size=in_size, in_height=in_size, in_width=in_size,
        in_voxel_size=dict(x=float(patch_size), y=float(patch_size), z=100.0),
        max_num_points_per_voxel=max_num_points_per_voxel, max_num_voxels=dict(train=g * g, test=g * g),
        out_feature_width=g, out_feature_height=g, out_feature_size=in_size,
        type=f"vit_small_patch{patch_size}_{in_size}.dino", checkpoint_file=None, pretrained=False,
        vit=dict(type=f"vit_small_patch{patch_size}_{in_size}.dino", checkpoint_file=None, pretrained=False,
                 depth=vit_depth, num_heads=vit_heads, mlp_dim=vit_mlp or 4 * patch_feature_dim),
        patch_size=patch_size, patch_feature_size=g, patch_feature_height=g, patch_feature_width=g,
        patch_feature_dim=patch_feature_dim, num_patches=g * g, out_feature_dim=out_feature_dim,
        image_mean=[0.0, 0.0, 0.0], image_std=[1.0, 1.0, 1.0], image_max_pixel_value=255.0,
    )
    mdl = dict(
        name=model, decoder=dict(in_feature_dim=out_feature_dim, in_feature_size=g),
        tokenizer=dict(num_bins=in_size, shuffle_tokens=False, max_num_vertices=max_num_vertices, max_len=None,
                       pad_idx=None, generation_steps=None),
        sinkhorn_iterations=sinkhorn_iterations, vertex_loss_weight=1.0, perm_loss_weight=10.0,
        batch_size=batch_size, learning_rate=3e-4, weight_decay=1e-4, num_epochs=200,
        compute_seg=True, compute_crossfield=True, seg=dict(compute_interior=True, compute_edge=False, compute_vertex=False),
        # config/model/ffl.yaml `loss:` (the FFL criterion, ffl_losses.build_combined_loss)
        loss=dict(multi=dict(epoch_thresholds=[0, 5, 10],
                             weights=dict(seg=1, crossfield_align=1, crossfield_align90=0.5, crossfield_smooth=0.005,
                                          seg_interior_crossfield=[0, 0, 0.2], seg_edge_crossfield=[0, 0, 0.2], seg_edge_interior=[0, 0, 0.2])),
                  seg=dict(bce_coef=1.0, dice_coef=0.2, use_freq=False, use_dist=False, use_size=False, w0=50, sigma=10, type="bool")),
    )
    return AttrDict.wrap(dict(
        experiment=dict(encoder=enc, model=mdl, lidar_dropout=lidar_dropout),
        host=dict(device=device, multi_gpu=multi_gpu),
        run_type=dict(name="release", logging="INFO", batch_size=batch_size),
        precision=precision,
    ))
