"""CPU restatement of the reference's refinement stage.  TEST INFRASTRUCTURE ONLY.

Follows /root/reference/tools/refinement.py: helpers :26-115 and the two-pass
``__main__`` block :135-428, on in-memory inputs.  Pinned by
tests/golden/refine_helpers_*.npz.  The CLIP text encoder is an input
(``encode_text``: str -> (1,D) tensor): the cosine formula :109-115 is pinned, the
absolute embedding values are parity unpinned (no CLIP weights offline).
"""
from __future__ import annotations

import torch

from .rle_ref import rle_decode_ref

# Reference idx_to_label :58-62 (class names keep their underscores).
SCANNET200 = (
    "chair.table.door.couch.cabinet.shelf.desk.office_chair.bed.pillow.sink.picture.window.toilet."
    "bookshelf.monitor.curtain.book.armchair.coffee_table.box.refrigerator.lamp.kitchen_cabinet.towel."
    "clothes.tv.nightstand.counter.dresser.stool.cushion.plant.ceiling.bathtub.end_table.dining_table."
    "keyboard.bag.backpack.toilet_paper.printer.tv_stand.whiteboard.blanket.shower_curtain.trash_can."
    "closet.stairs.microwave.stove.shoe.computer_tower.bottle.bin.ottoman.bench.board.washing_machine."
    "mirror.copier.basket.sofa_chair.file_cabinet.fan.laptop.shower.paper.person."
    "paper_towel_dispenser.oven.blinds.rack.plate.blackboard.piano.suitcase.rail.radiator."
    "recycling_bin.container.wardrobe.soap_dispenser.telephone.bucket.clock.stand.light."
    "laundry_basket.pipe.clothes_dryer.guitar.toilet_paper_holder.seat.speaker.column.bicycle.ladder."
    "bathroom_stall.shower_wall.cup.jacket.storage_bin.coffee_maker.dishwasher.paper_towel_roll."
    "machine.mat.windowsill.bar.toaster.bulletin_board.ironing_board.fireplace.soap_dish."
    "kitchen_counter.doorframe.toilet_paper_dispenser.mini_fridge.fire_extinguisher.ball.hat."
    "shower_curtain_rod.water_cooler.paper_cutter.tray.shower_door.pillar.ledge.toaster_oven.mouse."
    "toilet_seat_cover_dispenser.furniture.cart.storage_container.scale.tissue_box.light_switch.crate."
    "power_outlet.decoration.sign.projector.closet_door.vacuum_cleaner.candle.plunger.stuffed_animal."
    "headphones.dish_rack.broom.guitar_case.range_hood.dustpan.hair_dryer.water_bottle.handicap_bar."
    "purse.vent.shower_floor.water_pitcher.mailbox.bowl.paper_bag.alarm_clock.music_stand."
    "projector_screen.divider.laundry_detergent.bathroom_counter.object.bathroom_vanity.closet_wall."
    "laundry_hamper.bathroom_stall_door.ceiling_light.trash_bin.dumbbell.stair_rail.tube."
    "bathroom_cabinet.cd_case.closet_rod.coffee_kettle.structure.shower_head.keyboard_piano."
    "case_of_water_bottles.coat_rack.storage_organizer.folded_chair.fire_alarm.power_strip.calendar."
    "poster.potted_plant.luggage.mattress"
).split(".")


def idx_to_label_ref(idx) -> str:
    return SCANNET200[idx]


def iou_between_stages(mask_1: torch.Tensor, mask_2: torch.Tensor) -> torch.Tensor:
    """(n,x),(m,x) -> f32 (m,n).  Reference :69-90."""
    a, b = mask_1.float(), mask_2.float()
    inter = a @ b.T
    union = a.sum(dim=-1, keepdim=True) + b.sum(dim=-1, keepdim=True).T - inter
    return (inter / union).T


def text_cosine(encode_text, text1: str, text2: str) -> float:
    """cos of two text embeddings as a python float.  Reference :93-115."""
    with torch.no_grad():
        f1 = encode_text(text1)
        f2 = encode_text(text2)
    sim = f1 @ f2.T
    sim = sim / (f1.norm(dim=-1, keepdim=True) * f2.norm(dim=-1, keepdim=True).T)
    return sim.item()


def refine_class_ref(scenes, cfg, text_prompt: str, encode_text, return_debug: bool = False):
    """Reference :135-428.

    ``scenes``: list of (scene_id, stage1_dict_or_None, stage2_dict_or_None) in the sorted
    order of the stage-2 directory listing (:154); ``None`` models a missing file (:175-178).
    Returns {scene_id: final_output dict} for every scene the reference would save."""
    query_us = text_prompt.replace(" ", "_")                                   # :142
    all_ious, all_sims, all_m1, all_m2, all_conf2, all_other = [], [], [], [], [], []
    dbg = {"scenes": {}}

    for scene_id, stage1_in, stage2_in in scenes:                              # :166
        if stage1_in is None or stage2_in is None:                             # :175-178
            continue
        stage1 = dict(stage1_in)
        stage2 = dict(stage2_in)
        stage1["ins"] = torch.stack([torch.tensor(rle_decode_ref(r)) for r in stage1["ins"]])  # :187
        stage1["final_class"] = [idx_to_label_ref(i) for i in stage1["final_class"]]          # :193

        if len(stage2["conf"]) == 0:                                           # :196-205
            all_ious.append([]); all_sims.append([]); all_m1.append([])
            all_m2.append([]); all_conf2.append([])
            other = [i for i, lab in enumerate(stage1["final_class"]) if lab == query_us]
            all_other.append(stage1["ins"][other])
            continue

        iou = iou_between_stages(stage1["ins"], stage2["ins"])                 # :208
        best = torch.argmax(iou, dim=1)                                        # :211
        m_iou = iou_between_stages(stage1["ins"][best], stage1["ins"][best])   # :217
        m_iou[range(len(best)), range(len(best))] = 0                          # :221
        m_adj = (m_iou > cfg.stage1_iou_thres).to(int)                         # :224

        chosen = []                                                            # :230-249
        absorbed_by = torch.ones(len(m_adj), dtype=torch.int) * -1
        for i in range(len(m_adj)):
            if absorbed_by[i] != -1:
                chosen.append(best[absorbed_by[i]])
                continue
            chosen.append(best[i])
            if m_adj[i].sum() > 0:
                for j in range(len(m_adj[i])):
                    if m_adj[i][j] == 1:
                        absorbed_by[j] = i
                        stage1["ins"][best[i]] = stage1["ins"][best[i]] | stage1["ins"][best[j]]
        chosen = torch.tensor(chosen)                                          # :258
        uniq, cnt = torch.unique(chosen, return_counts=True)                   # :259
        for u, c in zip(uniq, cnt):                                            # :263-281
            if c > 1:
                sel = chosen == u
                merged = stage2["ins"][sel].any(dim=0)
                mconf = stage2["conf"][sel].mean()
                stage2["ins"] = torch.cat([stage2["ins"][~sel], merged.unsqueeze(0)])
                stage2["conf"] = torch.cat([stage2["conf"][~sel], mconf.unsqueeze(0)])
                chosen = torch.cat([chosen[~sel], u.unsqueeze(0)])

        iou = iou_between_stages(stage1["ins"], stage2["ins"])                 # :285
        best = torch.argmax(iou, dim=1)                                        # :288
        other = [i for i, lab in enumerate(stage1["final_class"])
                 if lab == query_us and i not in best]                         # :293
        all_other.append(stage1["ins"][other])
        labels = [stage1["final_class"][k] for k in best]                      # :297
        sims = [float(text_cosine(encode_text, text_prompt, lab)) for lab in labels]  # :299-302

        all_ious.append(iou[range(len(best)), best])                           # :308-312
        all_sims.append(sims)
        all_m1.append(stage1["ins"][best])
        all_m2.append(stage2["ins"])
        all_conf2.append(stage2["conf"])
        dbg["scenes"][scene_id] = {"best": best, "labels": labels, "sims": sims,
                                   "iou": iou[range(len(best)), best]}

    sim_unique = sorted(set(s for sims in all_sims for s in sims))             # :321-322
    sim_thres = sim_unique[int(len(sim_unique) * cfg.refinment_sim_percentile)]  # :324
    dbg["sim_unique"], dbg["sim_thres"] = sim_unique, sim_thres

    results = {}
    for s, (scene_id, _s1, _s2) in enumerate(scenes):                          # :330
        final = {"ins": [], "conf": [], "final_class": []}
        for row in all_other[s]:                                               # :340-343
            final["ins"].append(row)
            final["conf"].append(torch.tensor(0.5))
            final["final_class"].append(text_prompt)
        ious = all_ious[s]
        if len(ious) == 0:                                                     # :348-358
            if len(final["ins"]) != 0:
                final["ins"] = torch.stack(final["ins"]).to(bool)
                final["conf"] = torch.stack(final["conf"])
            results[scene_id] = final
            continue
        for m, v in enumerate(ious):                                           # :360-392
            if v > cfg.refiment_iou_thres:
                if all_sims[s][m] < sim_thres:
                    continue
                final["ins"].append(all_m1[s][m])
            else:
                final["ins"].append(all_m2[s][m])
            final["conf"].append(all_conf2[s][m])
            final["final_class"].append(text_prompt)
        if len(final["ins"]) == 0:                                             # :402-409
            results[scene_id] = final
            continue
        final["ins"] = torch.stack(final["ins"]).to(bool)                      # :411
        final["conf"] = torch.stack(final["conf"])                             # :412
        results[scene_id] = final
    return (results, dbg) if return_debug else results
