"""ScanNet200 class-name table used to turn Open3DIS stage-1 class indices into label strings.

Data table; order is the contract (reference tools/refinement.py:58-62, idx_to_label; the
reference string holds 198 names).  Names keep their underscores; the query class has spaces
replaced by underscores before comparison (:142).  Checked against the reference through
tests/golden/scannet200_labels.json.
"""

SCANNET200_LABELS = [
    'chair', 'table', 'door', 'couch', 'cabinet', 'shelf',
    'desk', 'office_chair', 'bed', 'pillow', 'sink', 'picture',
    'window', 'toilet', 'bookshelf', 'monitor', 'curtain', 'book',
    'armchair', 'coffee_table', 'box', 'refrigerator', 'lamp', 'kitchen_cabinet',
    'towel', 'clothes', 'tv', 'nightstand', 'counter', 'dresser',
    'stool', 'cushion', 'plant', 'ceiling', 'bathtub', 'end_table',
    'dining_table', 'keyboard', 'bag', 'backpack', 'toilet_paper', 'printer',
    'tv_stand', 'whiteboard', 'blanket', 'shower_curtain', 'trash_can', 'closet',
    'stairs', 'microwave', 'stove', 'shoe', 'computer_tower', 'bottle',
    'bin', 'ottoman', 'bench', 'board', 'washing_machine', 'mirror',
    'copier', 'basket', 'sofa_chair', 'file_cabinet', 'fan', 'laptop',
    'shower', 'paper', 'person', 'paper_towel_dispenser', 'oven', 'blinds',
    'rack', 'plate', 'blackboard', 'piano', 'suitcase', 'rail',
    'radiator', 'recycling_bin', 'container', 'wardrobe', 'soap_dispenser', 'telephone',
    'bucket', 'clock', 'stand', 'light', 'laundry_basket', 'pipe',
    'clothes_dryer', 'guitar', 'toilet_paper_holder', 'seat', 'speaker', 'column',
    'bicycle', 'ladder', 'bathroom_stall', 'shower_wall', 'cup', 'jacket',
    'storage_bin', 'coffee_maker', 'dishwasher', 'paper_towel_roll', 'machine', 'mat',
    'windowsill', 'bar', 'toaster', 'bulletin_board', 'ironing_board', 'fireplace',
    'soap_dish', 'kitchen_counter', 'doorframe', 'toilet_paper_dispenser', 'mini_fridge', 'fire_extinguisher',
    'ball', 'hat', 'shower_curtain_rod', 'water_cooler', 'paper_cutter', 'tray',
    'shower_door', 'pillar', 'ledge', 'toaster_oven', 'mouse', 'toilet_seat_cover_dispenser',
    'furniture', 'cart', 'storage_container', 'scale', 'tissue_box', 'light_switch',
    'crate', 'power_outlet', 'decoration', 'sign', 'projector', 'closet_door',
    'vacuum_cleaner', 'candle', 'plunger', 'stuffed_animal', 'headphones', 'dish_rack',
    'broom', 'guitar_case', 'range_hood', 'dustpan', 'hair_dryer', 'water_bottle',
    'handicap_bar', 'purse', 'vent', 'shower_floor', 'water_pitcher', 'mailbox',
    'bowl', 'paper_bag', 'alarm_clock', 'music_stand', 'projector_screen', 'divider',
    'laundry_detergent', 'bathroom_counter', 'object', 'bathroom_vanity', 'closet_wall', 'laundry_hamper',
    'bathroom_stall_door', 'ceiling_light', 'trash_bin', 'dumbbell', 'stair_rail', 'tube',
    'bathroom_cabinet', 'cd_case', 'closet_rod', 'coffee_kettle', 'structure', 'shower_head',
    'keyboard_piano', 'case_of_water_bottles', 'coat_rack', 'storage_organizer', 'folded_chair', 'fire_alarm',
    'power_strip', 'calendar', 'poster', 'potted_plant', 'luggage', 'mattress',
]


def idx_to_label(idx: int) -> str:
    return SCANNET200_LABELS[idx]
