# PatchRefinerPlus, DAv2 ViT-L coarse branch, MobileNetV4-small refiner, BiDirectionalFusion --
# the model dict of the reference's configs/patchrefinerv2_dav2/plus_mobile_u4k_base_coarse_e2e_c2f_pretrain.py
# (inference-relevant keys only; optimiser / dataloader sections are out of scope).
min_depth = 1e-3
max_depth = 80

model = dict(
    type='PatchRefinerPlus',
    config=dict(
        e2e_training=True, pretrain_stage=False,
        image_raw_shape=[2160, 3840], patch_process_shape=[448, 448], patch_raw_shape=[540, 960], patch_split_num=[4, 4],
        fusion_feat_level=6, min_depth=1e-3, max_depth=80,
        pretrain_coarse_model=None, strategy_refiner_target='offset_coarse',
        coarse_branch=dict(type='DA2', pretrained=None,
                           model_cfg=dict(encoder='vitl', features=256, out_channels=[256, 512, 1024, 1024])),
        refiner=dict(
            fine_branch=dict(type='LightWeightRefiner', coarse_condition=True, with_decoder=False,
                             encoder_name='mobilenetv4_conv_small.e2400_r224_in1k'),
            fusion_model=dict(type='BiDirectionalFusion', encoder_name='mobilenetv4_conv_small.e2400_r224_in1k',
                              coarse2fine=True, coarse2fine_type='coarse-gated',
                              coarse_chl=[128, 256, 256, 256, 256, 256], fine_chl=[32, 32, 64, 96, 960],
                              fine_chl_after_coarse2fine=[128, 256, 256, 256, 256, 256],
                              temp_chl=[32, 64, 64, 128, 256, 512], dec_chl=[512, 256, 128, 64, 32])),
        sigloss=dict(type='SILogLoss'), gmloss=dict(type='GradMatchLoss'), sigweight=1, pre_norm_bbox=True,
        pretrained=None, whole_pretrained=None))

general_dataloader = dict(
    batch_size=1, num_workers=2,
    dataset=dict(type='ImageDataset', rgb_image_dir='', dataset_name='', gt_dir=None,
                 network_process_size=(448, 448), resize_mode='depth-anything'))

collect_input_args = ['image_lr', 'image_hr', 'crops_image_hr', 'depth_gt', 'crop_depths', 'bboxs']
