"""transferattack_amd -- MI355X-native engine for the iterative FGSM-family hot path of
Trustworthy-AI-Group/TransferAttack, behind the reference's plug-in API:

    attack_zoo / load_attack_class      transferattack/__init__.py:3-160
    Attack and its hooks                transferattack/attack.py:8-169

Only the path BASELINE.json names is implemented (gradient/, input_transformation/, ensemble/ members in
the zoo below); other reference attacks ride on the same hooks but are not shipped here.
"""
import importlib

attack_zoo = {
    # gradient
    'fgsm': ('.gradient.fgsm', 'FGSM'),
    'ifgsm': ('.gradient.ifgsm', 'IFGSM'),
    'mifgsm': ('.gradient.mifgsm', 'MIFGSM'),
    'nifgsm': ('.gradient.nifgsm', 'NIFGSM'),
    'vmifgsm': ('.gradient.vmifgsm', 'VMIFGSM'),
    'vnifgsm': ('.gradient.vnifgsm', 'VNIFGSM'),
    'pifgsm': ('.gradient.pifgsm', 'PIFGSM'),
    'emifgsm': ('.gradient.emifgsm', 'EMIFGSM'),
    'iefgsm': ('.gradient.iefgsm', 'IEFGSM'),
    'gra': ('.gradient.gra', 'GRA'),
    'gnp': ('.gradient.gnp', 'GNP'),
    'pgn': ('.gradient.pgn', 'PGN'),
    'gifgsm': ('.gradient.gifgsm', 'GIFGSM'),
    'dta': ('.gradient.dta', 'DTA'),
    'pcifgsm': ('.gradient.pcifgsm', 'PCIFGSM'),
    'smifgrm': ('.gradient.smifgrm', 'SMIFGRM'),
    'fgsra': ('.gradient.fgsra', 'FGSRA'),
    'mig': ('.gradient.mig', 'MIG'),
    'aifgtm': ('.gradient.aifgtm', 'AIFGTM'),
    'mef': ('.gradient.mef', 'MEF'),
    'gaa': ('.gradient.gaa', 'GAA'),
    'ifgssm': ('.gradient.ifgssm', 'IFGSSM'),
    'vaifgsm': ('.gradient.vaifgsm', 'VAIFGSM'),
    'adamsi_fgm': ('.gradient.adamsi_fgm', 'AdaMSI_FGM'),
    'rgmifgsm': ('.gradient.mifgsm_with_tricks', 'RGMIFGSM'),
    'dual_mifgsm': ('.gradient.mifgsm_with_tricks', 'DualMIFGSM'),
    'ens_mifgsm': ('.gradient.mifgsm_with_tricks', 'Ens_FGSM_MIFGSM'),
    'anda': ('.gradient.anda', 'ANDA'),
    'rap': ('.gradient.rap', 'RAP'),
    'foolmix': ('.gradient.foolmix', 'Foolmix'),
    # input transformation
    'dim': ('.input_transformation.dim', 'DIM'),
    'tim': ('.input_transformation.tim', 'TIM'),
    'sim': ('.input_transformation.sim', 'SIM'),
    'admix': ('.input_transformation.admix', 'Admix'),
    'sia': ('.input_transformation.sia', 'SIA'),
    'bsr': ('.input_transformation.bsr', 'BSR'),
    'dem': ('.input_transformation.dem', 'DEM'),
    'ssm': ('.input_transformation.ssm', 'SSM'),
    'ssm_h': ('.input_transformation.ssm_with_tricks', 'SSM_H'),
    'ssm_p': ('.input_transformation.ssm_with_tricks', 'SSM_P'),
    'decowa': ('.input_transformation.decowa', 'DeCowA'),
    'ops': ('.input_transformation.ops', 'OPS'),
    'l2t': ('.input_transformation.l2t', 'L2T'),
    'su': ('.input_transformation.su', 'SU'),
    'everywhere': ('.input_transformation.everywhere', 'EverywhereAttack'),
    'maskblock': ('.input_transformation.maskblock', 'MaskBlock'),
    'usmm': ('.input_transformation.usmm', 'USMM'),
    'dts': ('.input_transformation.dts', 'DTS'),            # DIM+TIM+SIM composition (not in the reference zoo)
    # ensemble
    'ens': ('.ensemble.ens', 'ENS'),
    'svre': ('.ensemble.svre', 'SVRE'),
    'cwa': ('.ensemble.cwa', 'CWA'),
    'adaea': ('.ensemble.adaea', 'AdaEA'),
    'smer': ('.ensemble.smer', 'SMER'),
}


# Attacks whose result for one image does not depend on the other images of its batch (no shared geometry draw, no mixing
# partner, no batch statistic): the loop normalises the gradient per image (attack.py:124-128) and steps by its sign, so
# even the 1/N of the batch-mean loss drops out.  main.py may run several reference batches of these per device batch.
BATCH_INDEPENDENT = frozenset(['fgsm', 'ifgsm', 'mifgsm', 'nifgsm', 'vmifgsm', 'vnifgsm', 'tim', 'sim'])
# ... of which these draw noise on the device (in-kernel Philox over the flat element index of the device batch): their
# draws for one image depend on where the image sits in the device batch, so main.py does not regroup them by default
DEVICE_NOISE = frozenset(['vmifgsm', 'vnifgsm'])


def load_attack_class(attack_name):
    if attack_name not in attack_zoo:
        raise Exception('Unspported attack algorithm {}'.format(attack_name))
    module_path, class_name = attack_zoo[attack_name]
    module = importlib.import_module(module_path, __package__)
    return getattr(module, class_name)


__version__ = '0.1.0'
